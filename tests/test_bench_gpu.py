"""bench.py's N > 1 path (whole-clip sharding + the per-step all-gather of PredictionGatherer + max-over-ranks timing)
run for real: two ranks as fresh child processes of `torch.distributed.run`, both on GPU 0, gloo instead of RCCL
(BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist for exactly this).  The 8-GPU RCCL run is the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_gloo_on_one_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-roofline", "--clips-per-step", "4"],
             {"BENCH_DIST_BACKEND": "gloo", "BENCH_FORCE_DEVICE": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["value"] == d["value"] and d["value"] != float("inf")
    assert abs(d["value"] - 2 * 4 * 16 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job frames / max-rank time
    assert d["allgather_ms"] > 0 and d["allgather_bytes_per_rank"] == 4 * 176 * 26 * 4
    assert "scaling_note" in d and "rank(s)" in d["config"]["sharding"]
    w = d["workloads"]["64x36"]                                    # north_star's scaling workload rides along
    assert w["value"] > 0 and w["config"]["frames_per_clip"] == 64 and w["allgather_ms"] > 0


def test_bench_default_line_shape():
    """one rank, reduced step counts: the fields the driver and the judge read, incl. the per-kernel roofline rows and
    the 64x36 block measured in the same run"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "one_clip_per_pass", "workloads"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["by_kernel"] and r["by_shape"]
    gemm_rows = [x for x in r["by_kernel"] if x["class"] == "gemm"]
    # the GEMM class of the line is the sum of its per-kernel rows
    ms = sum(x["ms_per_step"] for x in gemm_rows)
    assert abs(ms - r["per_class_ms_per_step"]["gemm"]) < 1e-6 * max(ms, 1.0)
    gf = sum(x["gflop_per_step"] for x in gemm_rows)
    assert abs(gf / ms - r["achieved"]) < 1e-6 * r["achieved"]        # GFLOP per ms = TFLOP/s
    w = d["workloads"]["64x36"]
    assert w["value"] > 0 and w["roofline"]["frac"] > 0 and w["roofline"]["by_kernel"]
    # BASELINE configs[4] (DSG-DETR) and the configs[2] stand-in ride in the same line
    g = d["workloads"]["dsgdetr_16x12"]
    assert "error" not in g and g["value"] > 0 and g["roofline"]["frac"] > 0
    a = d["workloads"]["ag_split_shaped"]
    assert "error" not in a and a["value"] > 0 and a["clips"] == 256 and a["frames"] > 5000
    # --profile-only-batch: nothing but warm-up + timed steps
    p = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--profile-only-batch"])
    assert "roofline" not in p and "workloads" not in p and "one_clip_per_pass" not in p and "cpu_baseline" not in p
