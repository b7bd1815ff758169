#!/usr/bin/env python3
"""Where does the bf16x3 emulation stop being finite / accurate?  Sweeps the operand scales of a [600,256,1024] GEMM."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nl_vsgg_amd import _native  # noqa: E402

lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
M, N, K = 600, 256, 1024
g = torch.Generator(device="cuda").manual_seed(1)
A0 = torch.randn(M + 1, K, device="cuda", generator=g); W0 = torch.randn(N, K, device="cuda", generator=g)
for sa, sw in [(1e20, 1e-30), (1e10, 1e-10), (1e15, 1e-15), (1e18, 1e-18), (1e20, 1e-20), (1e19, 1), (1e20, 1), (1, 1e20), (1e-30, 1e20),
               (3e18, 1e-30), (1e19, 1e-30), (1e17, 1e-30)]:
    A = (A0 * sa).contiguous(); W = (W0 * sw).contiguous()
    C3 = torch.empty(M, N, device="cuda"); C1 = torch.empty(M, N, device="cuda")
    lib.sttran_debug_gemm_emulated(p(A), K, None, p(W), K, None, None, p(C3), M, N, K, 0, None)
    lib.sttran_debug_gemm_padded(p(A), K, None, p(W), K, None, None, p(C1), M, N, K, 0, 0, None)
    torch.cuda.synchronize()
    ref = A[:M].double() @ W.double().T
    mag = A[:M].double().abs() @ W.double().abs().T
    f3, f1 = bool(torch.isfinite(C3).all()), bool(torch.isfinite(C1).all())
    e3 = float(((C3.double() - ref).abs() / mag).nan_to_num(9.0).max()); e1 = float(((C1.double() - ref).abs() / mag).nan_to_num(9.0).max())
    print(f"a {sa:7.0e} w {sw:7.0e}: x3 finite {f3} err {e3:.2e} | exact finite {f1} err {e1:.2e}")
