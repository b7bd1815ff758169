#!/bin/bash
# same-box A/B of the whole bf16x3 step: current library vs tools/experiments/prev/libsttran_hip_prev.so (not committed)
mkdir -p gpurun_out
python -m pytest tests/test_sttran_gpu.py -m gpu -x -q -k "bf16x3 or golden" 2>&1 | tail -3
for L in "" tools/experiments/prev/libsttran_hip_prev.so; do
  STTRAN_LIB=$L BENCH_DETAIL=gpurun_out/r6f_detail.json python3 bench.py --steps 20 --no-strong --no-rccl-selftest --no-pcie --no-cpu-baseline > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
  python3 - "$L" <<'P'
import json, sys
d = json.load(open("gpurun_out/r6f_detail.json"))
print("LIB=%r" % sys.argv[1], d["value"])
for wl in ("16x12_bf16x3", "64x36_bf16x3"):
    w = d["workloads"][wl]
    print(wl, w.get("value"), w.get("error"), w.get("max_abs_diff_vs_fp32_engine"), w.get("roofline", {}).get("frac"))
    for r in w.get("roofline", {}).get("by_kernel", [])[:7]:
        print("   %-80s %5.1f x %8.1f us %6.1f TF ms=%.3f" % (r["kernel"][:80], r["launches_per_step"], r["mean_us"], r.get("tflops", 0), r["ms_per_step"]))
P
done
