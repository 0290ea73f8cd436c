"""Register / scratch / occupancy table of one translation unit's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py lfpsqp.jl_amd/csrc/retract.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *os.environ.get("KR_FLAGS", "").split(), f"-I{root}/include", "-c", src, "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if filt in r["name"]:
        print(f'{r.get("VGPRs",0):4d} v {r.get("AGPRs",0):4d} a  scratch {r.get("ScratchSize",0):5d}  spill {r.get("VGPRs Spill",0):4d}  occ {r.get("Occupancy",0)}  lds {r.get("LDS Size",0):6d}  {r["name"][:110]}')
