"""Secondary lines: the same workloads with the bf16x3 GEMM engine ("fp32 MFMA ... or 3 x bf16 split")."""
import torch

from nl_vsgg_amd.lib.sttran import pack_clips

from .common import BF16X3_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, device_clip
from .legs_batch import run_workload


def bf16x3_block(env, model, args, workload, cps, T, N):
    """The same workload with the bf16x3 GEMM engine (SURVEY.md 7's other admissible engine), and how far its outputs are
    from the exact engine's on the same batch.  A secondary line: never `value`, never `dtype`."""
    gen = torch.Generator(device=env.device).manual_seed(99)
    probe = [device_clip(T, N, gen, env.device) for _ in range(4 if workload == "16x12" else 1)]
    ref = {k: v.clone() for k, v in model(pack_clips(probe, copy=False)).items() if k.endswith("_distribution")}
    model.gemm_engine = "bf16x3"
    try:
        got = model(pack_clips(probe, copy=False))
        diff = max(float((got[k] - ref[k]).abs().max()) for k in ref)
        w = run_workload(env, model, args.model, workload, cps, max(5, min(args.steps, 20)), min(args.warmup, 3),
                         roofline=not args.no_roofline)
    finally:
        model.gemm_engine = "fp32"
    w.pop("unit", None)
    if "roofline" in w:
        # this engine's roof is the bf16 matrix pipe doing SIX bf16 products per fp32 product: 16 x the fp32-MFMA
        # rate / 6 (MI355X_MICROARCH.md: fp32 MFMA = 1/16 of bf16 MFMA), in fp32-equivalent TFLOP/s
        r = w["roofline"]
        r.pop("by_shape", None)
        r["peak"] = BF16X3_PEAK_TFLOPS
        r["frac"] = r["achieved"] / BF16X3_PEAK_TFLOPS
        r["unit"] = "TFLOP/s (fp32-equivalent: 2*M*N*K per launch)"
        r["kernel"] = ("the bf16x3 GEMM kernels + fix-up (bf16 MFMA, three bf16 planes per operand, six cross products) and the "
                       "launches that stay on the exact engine")
        for row in r.get("by_kernel", []):
            if "tflops" in row:
                row["frac_of_peak"] = row["tflops"] / (BF16X3_PEAK_TFLOPS if "x3" in row["kernel"] else FP32_MFMA_PEAK_TFLOPS)
        r["by_kernel_note"] = ("frac_of_peak of the x3 kernels' rows vs 419.5 TFLOP/s-equivalent (bf16 dense peak / 6), of the "
                               "other rows vs the fp32-MFMA peak 157.3")
    w["max_abs_diff_vs_fp32_engine"] = diff
    w["note"] = ("opt-in (model.gemm_engine = 'bf16x3'): nn.Linear GEMMs with M >= 512, the union 1x1 conv and the conv3x3 on the "
                 "bf16 matrix pipe, each fp32 operand split into three bf16 planes, six cross products, fp32 accumulate; error vs "
                 "fp64 no larger than the exact fp32-MFMA engine's (tests/test_kernels_gpu.py)")
    w.pop("reference_arithmetic", None)          # priced against the fp32 pipe: meaningless for this engine
    return w
