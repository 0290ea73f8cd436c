"""bench.py's harness logic (argument handling, JSON contract, roofline / cpu_baseline
objects) exercised at toy size on the emulator build; the numbers mean nothing here."""
import json

import bench


def test_bench_json_contract(emu_lib, capsys):
    out = bench.main(["--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "4096", "--cols", "8", "--cpu-sample-n", "4096"], lib=emu_lib)
    line = capsys.readouterr().out.strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert out["value"] > 0
