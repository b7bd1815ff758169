#!/bin/bash
# same-box A/B (ENGINE=bf16x3 for the second engine): the fused pair-conv kernel (default build) against -DSTTRAN_NO_CONV_FUSION (tools/experiments/build_variant.sh nofuse api_forward.hip -DSTTRAN_NO_CONV_FUSION)
mkdir -p gpurun_out
for rep in 1 2; do
for L in "" nl-vsgg_amd/csrc/ab/libsttran_hip_nofuse.so; do
  for wl in 16x12 64x36; do
    STTRAN_LIB=$L BENCH_DETAIL=gpurun_out/ab_fuse_detail.json python3 bench.py --workload $wl ${ENGINE:+--gemm-engine $ENGINE} --steps 20 --no-extra-workloads --no-cpu-baseline --no-strong --no-rccl-selftest --no-pcie > /dev/null 2> gpurun_out/ab_fuse.err
    python3 - "$L" $wl <<'P'
import json, sys
d = json.load(open("gpurun_out/ab_fuse_detail.json"))
r = d["roofline"]
rows = [x for x in r["by_kernel"] if "conv" in x["kernel"].lower() and "mask_conv1" not in x["kernel"]]
print("LIB=%-45r %s value %.0f frac %.4f | " % (sys.argv[1], sys.argv[2], d["value"], r["frac"]) + " ; ".join("%s %.1fx%.0fus" % (x["kernel"][:34], x["launches_per_step"], x["mean_us"]) for x in rows))
P
  done
done
done
