#!/usr/bin/env python3
"""Times the attention core (sttran_debug_attention) on the sequence shapes of the path:
    python tools/experiments/attn_bench.py
HBM floor = (qkv read + out written) / 8 TB/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nl_vsgg_amd import _native  # noqa: E402

lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
D, H = 1936, 8
for name, nseq, L in (("enc 16x12 x64", 1024, 11), ("dec 16x12 x64", 960, 22), ("dec 16x12 x1", 15, 22), ("enc 64x36 x4", 256, 35),
                      ("dec 64x36 x4", 252, 70), ("len 32", 660, 32), ("len 33", 640, 33)):
    tokens = nseq * L
    qkv = torch.randn(tokens, 3 * D, device="cuda")
    out = torch.empty(tokens, D, device="cuda")
    off = (torch.arange(nseq, device="cuda", dtype=torch.int32) * L).contiguous()
    ln = torch.full((nseq,), L, device="cuda", dtype=torch.int32)
    for _ in range(3):
        assert lib.sttran_debug_attention(p(qkv), p(off), p(ln), nseq, L, p(out), tokens, D, H, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.sttran_debug_attention(p(qkv), p(off), p(ln), nseq, L, p(out), tokens, D, H, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    byt = tokens * 4 * D * 4
    print(f"{name:16s} {nseq:5d} x {L:3d}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s  (floor {byt / 8e6:6.1f} us)")
