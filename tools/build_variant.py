"""Build a tuning variant of liblfpsqp_hip.so (extra -D flags) next to the product library, for A/B runs on the GPU box:
    python tools/build_variant.py <tag> -DLFPSQP_OP_LACC=0 ...   ->  lfpsqp.jl_amd/lib/variants/liblfpsqp_<tag>.so
    python bench.py --lib lfpsqp.jl_amd/lib/variants/liblfpsqp_<tag>.so ..."""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "lfpsqp.jl_amd", "lib", "variants")
obj = os.path.join(out, "obj_" + tag)
os.makedirs(obj, exist_ok=True)
hipcc = "/opt/rocm/bin/hipcc"
omp = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "lib")
srcs = sorted(glob.glob(os.path.join(ROOT, "lfpsqp.jl_amd", "csrc", "*.hip")))
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-fopenmp", "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma",
         "-I" + os.path.join(ROOT, "include"), *defs]
objs = [os.path.join(obj, os.path.basename(s)[:-4] + ".o") for s in srcs]
with ThreadPoolExecutor(8) as ex:
    list(ex.map(lambda so: subprocess.check_call([hipcc, *flags, "-c", so[0], "-o", so[1]]), zip(srcs, objs)))
lib = os.path.join(out, f"liblfpsqp_{tag}.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib, "-ldl", "-L" + omp, "-lomp", "-Wl,-rpath," + omp])
print(lib)
