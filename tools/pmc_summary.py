"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (MI355X_MICROARCH.md §HBM).
  16-byte-per-lane streaming kernels: bytes = FETCH_SIZE*1024*2 (gfx950 reports exactly half of a wide coalesced read).
  8-byte-per-lane kernels (onepass_kernel: 128-byte segments, 4 per wave instruction) are "uncalibrated" per the guide, so
  their factor is CALIBRATED on tools/micro/segprobe.hip's probe<4,16> -- the same access pattern reading a known
  n*m*8 bytes -- from an optional third FETCH_SIZE pass (argv[4]).
usage: pmc_summary.py FETCH_DIR WRITE_DIR OUT.json [CALIB_DIR]"""
import csv, glob, sys, json, collections
def load(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows
fd, wd = sys.argv[1], sys.argv[2]
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
factor8, calib = 2.0, None
if len(sys.argv) > 4:
    C = load(sys.argv[4], "FETCH_SIZE")
    for k, v in C.items():
        if "probe<4, 16>" in k or "probe<4,16>" in k:
            raw = sum(v) / len(v) * 1024.0
            known = 8.0 * 10000384 * 128
            factor8 = known / raw
            calib = dict(kernel=k, launches=len(v), fetch_raw_GB=raw / 1e9, known_GB=known / 1e9, factor=factor8)
out = {}
for k in sorted(F, key=lambda k: -sum(F[k])):
    f = sum(F[k]) / len(F[k]) * 1024.0
    w = (sum(W[k]) / len(W[k]) * 1024.0) if k in W else float('nan')
    fac = factor8 if "onepass_kernel" in k else 2.0
    out[k] = dict(launches=len(F[k]), fetch_raw_GB=f / 1e9, fetch_factor=fac, fetch_GB=fac * f / 1e9, write_GB=w / 1e9,
                  traffic_GB=(fac * f + w) / 1e9)
for k, v in list(out.items())[:14]:
    print(f"{v['traffic_GB']:9.3f} GB  (fetch raw {v['fetch_raw_GB']:7.3f} x{v['fetch_factor']:.3f} = {v['fetch_GB']:7.3f}  write {v['write_GB']:6.3f})  n={v['launches']:3d}  {k[:100]}")
if calib: print("calibration:", calib)
# checksum of the library these passes profiled (bench.py quotes `traffic` as measured only for the same build)
import hashlib, os
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lfpsqp.jl_amd", "lib", "liblfpsqp_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
json.dump(dict(kernels=out, calibration=calib, lib_sha256=sha), open(sys.argv[3], "w"), indent=1)
