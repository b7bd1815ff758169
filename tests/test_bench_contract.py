"""The bench line the driver parses: the committed end-of-round line (profiles/) must carry every field of the
bench.py contract, and bench.py must still know the flags the driver passes."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    lines = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("bench_default_with_cpu.json"))
    d = json.load(open(os.path.join(ROOT, "profiles", lines[-1])))          # the newest round's default line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] < 1.0 and r["peak"] == 157.3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["host_cores"] >= c["cores"]
    # round 2: per-kernel roofline rows and the second BASELINE workload ride in the same line
    assert r["by_kernel"] and all(k in r["by_kernel"][0] for k in ("kernel", "launches_per_step", "gflop_per_step", "mean_us"))
    w = d["workloads"]["64x36"]
    assert w["value"] > 0 and 0.0 < w["roofline"]["frac"] < 1.0 and w["config"]["frames_per_clip"] == 64
    assert d["one_clip_per_pass"]["value"] > 0
    assert abs(d["value"] - d["config"]["clips_per_step"] * d["config"]["frames_per_clip"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_accepts_the_driver_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(rf'add_argument\("{flag}"', src), flag
    assert 'all_reduce(t, op=self.dist.ReduceOp.MAX)' in src           # max over ranks
    assert "launch_ranks(args.gpus" in src and "os.exec" not in src     # --gpus N without a launcher: fresh children, never exec
    assert src.count("env.barrier(gatherer)") >= 2                      # both sides of the timed region
    assert "PredictionGatherer" in src                                  # the tested gather is the timed gather


def test_newest_driver_bench_line_has_the_contract_fields():
    """the DRIVER's own record of the last round (BENCH_rNN.json at the repo root: `parsed` = the line it read from
    bench.py on a fresh MI355X): the same contract checks as for the committed profile line"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")))
    if not files:
        import pytest
        pytest.skip("no driver bench record in this checkout")
    rec = json.load(open(files[-1]))
    d = rec.get("parsed") or {}
    assert rec.get("rc", 0) == 0 and d, files[-1]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (files[-1], k)
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0
    # value = whole-job frames / time of the timed steps
    per_step = d["config"]["clips_per_step"] * d["config"]["frames_per_clip"] * d["n_gpus"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
