#!/usr/bin/env python3
"""Phase clocks of the 128 x 176 GEMM tile (library built with EXTRA=-DSTTRAN_GEMM_EXPERIMENT): where a workgroup's time goes
between tile start, first MFMA, last MFMA and the end of the epilogue.  usage: t16_phases.py M,N,K [residual]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nl_vsgg_amd import _native  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1].split(","))
res = len(sys.argv) > 2
lib = _native.load()
raw = C.CDLL(_native.LIB_PATH)
p = lambda t: C.c_void_p(t.data_ptr())
Kp = (K + 31) // 32 * 32
A = torch.randn(M + 1, Kp, device="cuda"); W = torch.zeros(N, Kp, device="cuda"); W[:, :K] = torch.randn(N, K, device="cuda")
b = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda"); R = torch.randn(M, N, device="cuda") if res else None
os.environ["STTRAN_T16_ABLATE"] = "9"
for _ in range(30):
    lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), p(R) if res else None, p(Cc), M, N, K, 0, 5, None)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
raw.sttran_debug_t16_clocks(out)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), p(R) if res else None, p(Cc), M, N, K, 0, 5, None)
e1.record(); torch.cuda.synchronize()
raw.sttran_debug_t16_clocks(out)
us = e0.elapsed_time(e1) * 1e3 / n
pro, loop, epi, tiles, steps = (out[i] / n for i in range(5))
tot = pro + loop + epi
print(f"{M}x{N}x{K}: {us:.1f} us/launch, {2.0 * M * N * K / us / 1e6:.1f} TF; per launch: {tiles:.0f} tile segments, {steps:.0f} K-steps")
print(f"  share of workgroup time: prologue {pro / tot:.3f}  main loop {loop / tot:.3f}  epilogue/park {epi / tot:.3f}")
print(f"  per tile segment (ticks of s_memtime, 100 MHz => x10 ns): prologue {pro / tiles:.0f}  epilogue {epi / tiles:.0f}; "
      f"main loop per K-step {loop / steps:.1f}")
