#!/usr/bin/env python3
"""Run-to-run determinism of the second engine: (1) the 16x16x32 kernel hook on small / ragged row counts, repeated, compared
bitwise; (2) whole bf16x3_all forwards of one small clip, repeated on one handle and on fresh handles, compared bitwise and
against the exact engine.  STTRAN_GUARD_WORKSPACE=1 puts the library's buffers in guarded mappings (where the difference
showed first)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
from nl_vsgg_amd import _native  # noqa: E402
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
KEYS = ("attention_distribution", "spatial_distribution", "contacting_distribution")


def kernels():
    g = torch.Generator().manual_seed(3)
    for M, N, K in ((160, 1936, 1936), (160, 5808, 1936), (160, 2048, 1936), (160, 1936, 2048), (37, 1936, 1936), (500, 1936, 1936),
                    (1, 1936, 1936), (160, 1024, 2048), (160, 512, 1936)):
        Kp = (K + 31) // 32 * 32
        A = torch.zeros(M + 1, Kp); A[:, :K] = torch.randn(M + 1, K, generator=g)
        W = torch.zeros(N, Kp); W[:, :K] = torch.randn(N, K, generator=g)
        A, W = A.cuda(), W.cuda()
        b = torch.randn(N, generator=g).cuda()
        first, bad = None, 0
        for it in range(30):
            Cc = torch.full((M, N), float("nan"), device="cuda")
            rc = lib.sttran_debug_gemm_emulated_t16(p(A), Kp, None, p(W), Kp, p(b), None, p(Cc), M, N, K, 0, None)
            torch.cuda.synchronize()
            if rc != 0:
                print("rc", rc, M, N, K); break
            if first is None:
                first = Cc.clone()
                ref = A[:M, :K].double() @ W[:, :K].double().T + b.double()
                err = (Cc.double() - ref).abs().max().item()
            elif not torch.equal(first, Cc):
                bad += 1
        print("kernel M=%d N=%d K=%d: %d of 29 repeats differ, err vs fp64 %.2e" % (M, N, K, bad, err))


def forwards():
    import guarded_child as gc
    sd = syn.make_sttran_state_dict(7)
    e = syn.make_entry(22, [11] * 16, mode="predcls")
    e = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
    m = gc._model("predcls", sd)
    want = {k: v.clone() for k, v in m(dict(e)).items() if k in KEYS}
    m.sync_check()
    for eng in ("bf16x3_all", "bf16x3"):
        m.gemm_engine = eng
        first = None
        for it in range(12):
            out = m(dict(e)); m.sync_check()
            got = {k: out[k].clone() for k in KEYS}
            if first is None:
                first = got
                print(eng, "max |diff| vs exact engine:", {k: float((got[k] - want[k]).abs().max()) for k in KEYS})
            else:
                d = {k: float((got[k] - first[k]).abs().max()) for k in KEYS}
                if any(v != 0 for v in d.values()):
                    print(eng, "repeat", it, "differs from repeat 0:", d)
        print(eng, "same handle: done")
    firsts = []
    for it in range(6):
        m2 = gc._model("predcls", sd)
        m2.gemm_engine = "bf16x3_all"
        out = m2(dict(e)); m2.sync_check()
        firsts.append({k: out[k].clone() for k in KEYS})
        d = {k: float((firsts[-1][k] - firsts[0][k]).abs().max()) for k in KEYS}
        print("fresh handle", it, "vs fresh handle 0:", d, "vs exact:", {k: float((firsts[-1][k] - want[k]).abs().max()) for k in KEYS})
        del m2


def child_sequence():
    """the guarded-workspace child's own order: by-pointer batch on a fresh handle, engine switch, one clip"""
    import guarded_child as gc
    from nl_vsgg_amd.lib.sttran import pack_clips
    sd = syn.make_sttran_state_dict(7)
    clips = []
    for seed, counts in enumerate(([3, 1, 4, 2, 2], [11] * 16, [0, 2, 0, 3], [7] * 30)):
        e = syn.make_entry(seed + 21, counts, mode="predcls")
        clips.append({k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()})
    m = gc._model("predcls", sd)
    want = {k: v.clone() for k, v in m(dict(clips[1])).items() if k in KEYS}
    m.sync_check()
    for rep in range(2):
        m2 = gc._model("predcls", sd)
        out = m2(pack_clips([dict(c) for c in clips], copy=False)); m2.sync_check()
        m2.gemm_engine = "bf16x3_all"
        first = None
        for it in range(4):
            out = m2(dict(clips[1])); m2.sync_check()
            got = {k: out[k].clone() for k in KEYS}
            if first is None:
                first = got
            if it < 2:
                print("  handle %d forward %d vs exact: %s%s" % (rep, it, " ".join("%.2e" % float((got[k] - want[k]).abs().max()) for k in KEYS),
                      " NAN" if any(not bool(torch.isfinite(got[k]).all()) for k in KEYS) else ""))


if __name__ == "__main__":
    if "child" in sys.argv[1:]:
        child_sequence()
        sys.exit(0)
    if "forward" not in sys.argv[1:]:
        kernels()
    if "kernel" not in sys.argv[1:]:
        forwards()
