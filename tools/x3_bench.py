#!/usr/bin/env python3
"""Micro-benchmark of the second-generation bf16x3 GEMM (csrc/gemm_bf16x3_t16.h) per shape: the kernel alone on pre-split
operands, the activation split pass alone, round 2's kernel (in-loader split) and the exact fp32-MFMA engine beside them.
    python tools/x3_bench.py [--shapes path16x64|path64|big] [--one M,N,K] [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nl_vsgg_amd import _native  # noqa: E402
from gemm_bench import SHAPES  # noqa: E402

PEAK_X3 = 16 * 157.3 / 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="path16x64")
    ap.add_argument("--one", default="")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    os.environ["STTRAN_X3_CACHE_PLANES"] = "1"
    lib = _native.load()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    shapes = [("one",) + tuple(int(v) for v in a.one.split(","))] if a.one else SHAPES[a.shapes]
    for name, M, N, K in shapes:
        if N % 128 and N % 176:
            continue
        Kp = (K + 31) // 32 * 32
        A = torch.randn(M + 1, Kp, device="cuda"); A[:, K:] = 0
        W = torch.zeros(N, Kp, device="cuda"); W[:, :K] = torch.randn(N, K, device="cuda")
        b = torch.randn(N, device="cuda")
        Cc = torch.empty(M, N, device="cuda")
        us = (C.c_double * 2)()
        rc = lib.sttran_debug_x3t16_bench(p(A), Kp, p(W), Kp, p(b), None, p(Cc), M, N, K, a.iters, us)
        if rc != 0:
            print(f"{name}: rc {rc}")
            continue
        ref = A[:M, :K].double() @ W[:, :K].double().T + b.double()
        err = (Cc.double() - ref).abs().max().item()

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters
        us_old = timed(lambda: lib.sttran_debug_gemm_emulated(p(A), Kp, None, p(W), Kp, p(b), None, p(Cc), M, N, K, 0, None))
        us_f32 = timed(lambda: lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), None, p(Cc), M, N, K, 0, 0, None))
        fl = 2.0 * M * N * K / 1e6
        print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d} | x3t16 {us[0]:8.1f} us {fl / us[0]:6.1f} TF-eq ({fl / us[0] / PEAK_X3:.3f}) + split "
              f"{us[1]:6.1f} us -> {fl / (us[0] + us[1]):6.1f} | round-2 x3 {us_old:8.1f} us {fl / us_old:6.1f} | fp32 {us_f32:8.1f} us "
              f"{fl / us_f32:6.1f} | max err vs fp64 {err:.2e}")


if __name__ == "__main__":
    main()
