#!/bin/bash
# same-box A/B of the bf16x3 kernels: current library vs tools/experiments/prev/libsttran_hip_prev.so (not committed)
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "bf16x3 or gemm_tiles or padded" 2>&1 | tail -3
python -m pytest tests/test_sttran_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== new"; python tools/x3_bench.py --shapes path16x64 2>&1 | grep -v amdgpu.ids
echo "== prev"; STTRAN_LIB=tools/experiments/prev/libsttran_hip_prev.so python tools/x3_bench.py --shapes path16x64 2>&1 | grep -v amdgpu.ids | grep "dec_\|enc_"
echo "== gemm_bench vr_fc new/prev"
python tools/gemm_bench.py --one 11264,512,12544 --tiles 7 --iters 20 2>&1 | tail -2
STTRAN_LIB=tools/experiments/prev/libsttran_hip_prev.so python tools/gemm_bench.py --one 11264,512,12544 --tiles 7 --iters 20 2>&1 | tail -2
for L in "" tools/experiments/prev/libsttran_hip_prev.so; do
  STTRAN_LIB=$L BENCH_DETAIL=gpurun_out/r6e_detail.json python3 bench.py --steps 20 --no-strong --no-rccl-selftest --no-pcie --no-cpu-baseline > gpurun_out/r6e_bench.json 2> gpurun_out/r6e_bench.err
  python3 - "$L" <<'P'
import json, sys
d = json.load(open("gpurun_out/r6e_detail.json"))
print("LIB=%r" % sys.argv[1], d["value"], d["roofline"]["per_class_ms_per_step"])
for wl in ("16x12_bf16x3", "64x36_bf16x3"):
    w = d["workloads"][wl]
    print(wl, w.get("value"), w.get("error"), w.get("max_abs_diff_vs_fp32_engine"), w.get("roofline", {}).get("frac"))
    for r in w.get("roofline", {}).get("by_kernel", [])[:8]:
        print("   %-80s %5.1f x %8.1f us %6.1f TF ms=%.3f" % (r["kernel"][:80], r["launches_per_step"], r["mean_us"], r.get("tflops", 0), r["ms_per_step"]))
P
done
