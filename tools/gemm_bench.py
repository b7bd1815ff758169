#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM (sttran_debug_gemm_padded: the product's operand layout) per (shape, tile).
Used to tune the tile planner (csrc/kernels_gemm.hip); run on the GPU box:
    python tools/gemm_bench.py [--shapes big|path64|path16|path16x8|path16x16|path16x64] [--tiles 1,2,3,4,5]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nl_vsgg_amd import _native  # noqa: E402

TILES = {1: "256x128", 2: "128x128", 3: "64x64", 4: "128x64", 5: "128x176", 7: "T128x128"}


def path_shapes(P, NT):
    D, F = 1936, 2048
    return [("fc", P, 512, 2048), ("conv", 256, P * 49, 1152), ("vr_fc", P, 512, 12544),
            ("enc_qkv", P, 3 * D, D), ("enc_out", P, D, D), ("enc_ffn1", P, F, D), ("enc_ffn2", P, D, F),
            ("dec_qkv", NT, 3 * D, D), ("dec_out", NT, D, D), ("dec_ffn1", NT, F, D), ("dec_ffn2", NT, D, F),
            ("heads", P, 26, D)]


SHAPES = {
    "big": [("sq4096", 4096, 4096, 4096), ("sq8192x2048", 8192, 8192, 2048)],
    "path64": path_shapes(2240, 4410),
    "path16": path_shapes(176, 330),
    "path16x8": path_shapes(1408, 2640),
    "path16x16": path_shapes(2816, 5280),
    "path16x64": path_shapes(11264, 21120),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="big")
    ap.add_argument("--one", default="", help="M,N,K: a single shape instead of a --shapes set")
    ap.add_argument("--tiles", default="1,2,3,4")
    ap.add_argument("--pipes", default="0", help="main-loop variants (library built with EXTRA=-DSTTRAN_GEMM_EXPERIMENT)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--residual", action="store_true", help="add a residual [M,N] in the epilogue (out-proj / FFN2 form)")
    ap.add_argument("--x3", action="store_true", help="also time the bf16x3 fp32-emulation engine (experiment)")
    ap.add_argument("--zeros", action="store_true", help="zero operands: separates clock (power) limits from schedule limits")
    ap.add_argument("--cold", action="store_true",
                    help="overwrite a 768 MB buffer before every timed launch (operands come from HBM, not from the "
                         "256 MB Infinity Cache a back-to-back replay keeps warm); each launch timed on its own")
    a = ap.parse_args()
    os.environ["STTRAN_X3_CACHE_PLANES"] = "1"       # --x3: time the GEMM, not the weight split (W stays put here)
    lib = _native.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    torch.zeros(1, device="cuda")
    pk = C.c_double()
    for _ in range(3):
        lib.sttran_debug_mfma_peak(20000, C.byref(pk))
    print(f"device fp32-MFMA rate (register-only loop): {pk.value:.1f} TFLOP/s (spec peak 157.3)")
    flush = torch.zeros(192 * 1024 * 1024, device="cuda") if a.cold else None
    shapes = [("one",) + tuple(int(v) for v in a.one.split(","))] if a.one else SHAPES[a.shapes]
    for name, M, N, K in shapes:
        Kp = (K + 31) // 32 * 32                     # the product's layout: rows padded to 32 columns, W zero there
        A = torch.randn(M + 1, Kp, device="cuda")
        W = torch.zeros(N, Kp, device="cuda")
        W[:, :K] = torch.randn(N, K, device="cuda")
        if a.zeros:
            A.zero_(); W.zero_()
        b = torch.randn(N, device="cuda")
        Cc = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, device="cuda") if a.residual else None
        pr = p(R) if a.residual else None
        best = None
        rows = []
        # clocks ramp over the first tens of milliseconds of load after an idle gap (the allocations above): without this
        # the FIRST configuration measured for a shape reads 2-8 % low
        t_end = torch.cuda.Event(enable_timing=True); t_beg = torch.cuda.Event(enable_timing=True)
        t_beg.record()
        for _ in range(max(4, int(6e10 / max(1.0, 2.0 * M * N * K)))):       # ~60 ms of work
            lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), pr, p(Cc), M, N, K, 0, 1, None)
        t_end.record(); torch.cuda.synchronize()
        for tile in [0] + [int(t) for t in a.tiles.split(",")]:
            for split in ([0] if tile == 0 else [int(s) for s in a.pipes.split(",")]):
                os.environ["STTRAN_GEMM_PIPE"] = str(split)     # only read by EXPERIMENT builds
                os.environ["STTRAN_T16_ABLATE"] = str(split)    # ... the 128x176 tile's ablations (same --pipes list)
                for _ in range(2):
                    lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), pr, p(Cc), M, N, K, 0, tile, None)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if a.cold:
                    us = 0.0
                    for _ in range(a.iters):
                        flush.add_(1.0)
                        e0.record()
                        lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), pr, p(Cc), M, N, K, 0, tile, None)
                        e1.record()
                        torch.cuda.synchronize()
                        us += e0.elapsed_time(e1) * 1e3 / a.iters
                else:
                    e0.record()
                    for _ in range(a.iters):
                        lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), pr, p(Cc), M, N, K, 0, tile, None)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / a.iters
                tf = 2.0 * M * N * K / us / 1e6
                rows.append((tile, split, us, tf))
                if tile and (best is None or us < best[2]):
                    best = (tile, split, us, tf)
        if a.x3:
            Wd = W[:, :K].contiguous() if Kp != K else W
            for _ in range(3):
                lib.sttran_debug_gemm_emulated(p(A), Kp, None, p(Wd), Wd.shape[1], p(b), pr, p(Cc), M, N, K, 0, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                lib.sttran_debug_gemm_emulated(p(A), Kp, None, p(Wd), Wd.shape[1], p(b), pr, p(Cc), M, N, K, 0, None)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            ref = (A[:M, :K].double() @ Wd.double().T + b.double() + (R.double() if a.residual else 0))
            err3 = (Cc.double() - ref).abs().max().item()
            lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), pr, p(Cc), M, N, K, 0, 1, None)
            torch.cuda.synchronize()
            err1 = (Cc.double() - ref).abs().max().item()
            print(f"     bf16x3: {us:9.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF-equivalent   max|err| vs fp64: x3 {err3:.3e}  fp32-MFMA {err1:.3e}")
        auto = rows[0]
        print(f"{name:10s} M={M:6d} N={N:6d} K={K:5d}  auto[{TILES.get(lib.sttran_debug_plan_tile(M, N, K), '?')}] {auto[2]:9.1f} us {auto[3]:6.1f} TF | best "
              f"{TILES[best[0]]}/p{best[1]} {best[2]:9.1f} us {best[3]:6.1f} TF")
        print("     " + "  ".join(f"{TILES[t]}/p{s}:{tf:5.1f} ({us:.1f} us)" for t, s, us, tf in rows[1:]))


if __name__ == "__main__":
    main()
