"""Instruction mix of one kernel's ISA (cross-compiled for gfx950, no GPU needed): MFMA / vector / scalar / LDS / memory instructions in total and inside
the innermost loops, and the vector opcodes that occur most.  How FINDINGS.md 12.7 / 12.8 were found: a register array demoted to scratch shows up as
scratch_load / scratch_store, run-time null tests of kernel arguments as v_cmp / v_cndmask between the v_mfma.
    python tools/isa_mix.py lfpsqp.jl_amd/csrc/factorize.hip 'gram_kernel<true, false, false, 0>'"""
import collections, os, re, subprocess, sys, tempfile
src, want = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{root}/include", src, "--cuda-device-only", "-S", "-o", out],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s", l)]
names = {}
for i in starts:
    sym = lines[i].split(":")[0]
    names[i] = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
hits = [i for i in starts if want in names[i]]
if not hits:
    sys.exit("no kernel matches; candidates:\n  " + "\n  ".join(sorted(set(n[:140] for n in names.values() if "lfpsqp" in n))[:60]))
i0 = hits[0]
i1 = next(j for j in range(i0, len(lines)) if "s_endpgm" in lines[j])
body = lines[i0:i1 + 1]


def mix(ls):
    c = collections.Counter()
    ops = collections.Counter()
    for l in ls:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1; ops[op] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("scratch_"): c["SCRATCH"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
    return c, ops


c, ops = mix(body)
print(names[i0][:160])
print("whole kernel:", dict(c))
# loops: a label that a later branch jumps back to
labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for k, l in enumerate(body):
    m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < k:
            loops.append((labels[t], k))
for a, b in sorted(set(loops), key=lambda ab: ab[0] - ab[1])[:4]:
    c, ops = mix(body[a:b + 1])
    print(f"loop lines {a}..{b} ({b - a + 1} lines):", dict(c), "| top vector opcodes:", ", ".join(f"{v} {k}" for k, v in ops.most_common(6)))
