"""Child process of tests/test_guarded_buffers_gpu.py: runs GEMMs and whole forwards on operands that END at the end
of a device mapping (sttran_debug_guarded_alloc: the page behind is reserved, unmapped address space).  A kernel that
touches memory past a caller's buffer faults and the process aborts; the parent reports it.  Prints one `OK ...` line
per passed group.  Usage: python tests/helpers/guarded_child.py probe|gemm|forward|aux|workspace"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nl_vsgg_amd import _native  # noqa: E402
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

lib = _native.load()
_TYPESTR = {torch.float32: "<f4", torch.int64: "<i8", torch.int32: "<i4", torch.uint8: "|u1"}
_keep = []


class _Holder:
    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": _TYPESTR[dtype], "data": (ptr, False),
                                         "version": 2, "strides": None}


def guarded(t):
    """a copy of tensor `t` in guarded device memory (its last element is the last mapped one)"""
    t = t.contiguous()
    if t.numel() == 0:
        return t.cuda()
    nbytes = t.numel() * t.element_size()
    ptr, cookie = C.c_void_p(), C.c_void_p()
    rc = lib.sttran_debug_guarded_alloc(nbytes, C.byref(ptr), C.byref(cookie))
    if rc != 0:
        print("NOVMM rc=%d" % rc)
        sys.exit(0)
    span = (nbytes + 15) & ~15
    g = torch.as_tensor(_Holder(ptr.value + (span - nbytes), t.shape, t.dtype), device="cuda")   # end-aligned
    g.copy_(t)
    _keep.append((cookie, g))
    return g


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def step(*what):
    """the case about to run, on stderr: after a fault the last line names the culprit"""
    print("RUN", *what, file=sys.stderr, flush=True)


def run_gemm():
    gen = torch.Generator().manual_seed(5)
    cases = [(2240, 1936, 1936), (4410, 2048, 1936), (1100, 1936, 2048), (176, 1936, 1936), (330, 5808, 1936), (129, 352, 100),
             (1, 176, 32), (2240, 512, 2048), (300, 26, 1936)]
    n = 0
    for M, N, K in cases:
        Kp = (K + 31) // 32 * 32
        A = torch.randn(M, Kp, generator=gen)
        W = torch.zeros(N, Kp); W[:, :K] = torch.randn(N, K, generator=gen) * 0.05
        b, res = torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
        ref = (A[:, :K].double() @ W[:, :K].double().T + b.double()).clamp_min(0) + res.double()
        gA, gW, gb, gres = guarded(A), guarded(W), guarded(b), guarded(res)
        for tile in range(0, 8):
            if tile == 5 and N % 176:
                continue
            if tile == 7 and N % 128:
                continue
            step("padded", M, N, K, "tile", tile)
            gC = guarded(torch.full((M, N), float("nan")))
            rc = lib.sttran_debug_gemm_padded(p(gA), Kp, None, p(gW), Kp, p(gb), p(gres), p(gC), M, N, K, 1, tile, None)
            assert rc == 0, (M, N, K, tile, rc)
            torch.cuda.synchronize()
            err = (gC.cpu().double() - ref).abs().max().item()
            assert err < 2e-3, (M, N, K, tile, err)
            n += 1
    # gathered A rows (subj / obj FC, last-layer row pruning) and the bias-only / no-epilogue forms
    M, N, K, R = 700, 1936, 1936, 300
    Kp = (K + 31) // 32 * 32
    A = torch.randn(R, Kp, generator=gen)
    W = torch.zeros(N, Kp); W[:, :K] = torch.randn(N, K, generator=gen) * 0.05
    b = torch.randn(N, generator=gen)
    idx = torch.randint(0, R, (M,), generator=gen, dtype=torch.int32)
    ref = A[idx.long(), :K].double() @ W[:, :K].double().T
    gA, gW, gb, gidx = guarded(A), guarded(W), guarded(b), guarded(idx)
    for tile in range(0, 8):
        if tile == 7:
            continue
        for bias in (gb, None):
            step("gathered", M, N, K, "tile", tile, "bias", bias is not None)
            gC = guarded(torch.full((M, N), float("nan")))
            assert lib.sttran_debug_gemm_padded(p(gA), Kp, p(gidx), p(gW), Kp, p(bias), None, p(gC), M, N, K, 0, tile, None) == 0
            torch.cuda.synchronize()
            err = (gC.cpu().double() - ref - (b.double() if bias is not None else 0)).abs().max().item()
            assert err < 2e-3, ("gather", tile, err)
            n += 1
    # arbitrary operands (the zero-select loader): K is not a multiple of 32 and nothing is padded
    for M, N, K in ((33, 70, 100), (257, 129, 36), (300, 26, 1936), (200, 1024, 2376), (1, 3, 4)):
        A, W = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) * 0.05
        b, res = torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
        ref = A.double() @ W.double().T + b.double() + res.double()
        gA, gW, gb, gres = guarded(A), guarded(W), guarded(b), guarded(res)
        for tile in (0, 1, 2, 3, 4):
            step("select", M, N, K, "tile", tile)
            gC = guarded(torch.full((M, N), float("nan")))
            assert lib.sttran_debug_gemm(p(gA), None, p(gW), p(gb), p(gres), p(gC), M, N, K, 0, tile, 1, None) == 0
            torch.cuda.synchronize()
            err = (gC.cpu().double() - ref).abs().max().item()
            assert err < 2e-3, ("select", M, N, K, tile, err)
            n += 1
    # attention and LayerNorm on exactly-sized buffers
    D, H = 1936, 8
    # (round 5: launches that fill the chip take other variants of the short-sequence kernel -- 300 x 11, 256 x 22, 230 x 24)
    for nseq, L in ((15, 22), (16, 11), (3, 35), (3, 70), (2, 81), (1, 500), (5, 32), (5, 33), (1, 1537), (2, 481),
                    (300, 11), (256, 22), (230, 24), (40, 48), (40, 77)):
        tokens = nseq * L
        step("attention / layernorm", nseq, L)
        qkv = guarded(torch.randn(tokens, 3 * D, generator=gen))
        out = guarded(torch.full((tokens, D), float("nan")))
        off = guarded((torch.arange(nseq, dtype=torch.int32) * L).contiguous())
        ln = guarded(torch.full((nseq,), L, dtype=torch.int32))
        assert lib.sttran_debug_attention(p(qkv), p(off), p(ln), nseq, L, p(out), tokens, D, H, None) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        x, g, bt = guarded(torch.randn(tokens, D, generator=gen)), guarded(torch.rand(D, generator=gen)), guarded(torch.randn(D, generator=gen))
        y = guarded(torch.full((tokens, D), float("nan")))
        assert lib.sttran_debug_layernorm(p(x), p(g), p(bt), p(y), tokens, D, None) == 0
        torch.cuda.synchronize()
        want = torch.nn.functional.layer_norm(x.cpu().double(), (D,), g.cpu().double(), bt.cpu().double(), 1e-5)
        assert (y.cpu().double() - want).abs().max().item() < 1e-4
        n += 2
    print("OK gemm %d launches" % n)


CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]


def _model(mode, sd):
    from nl_vsgg_amd.lib.sttran import STTran
    m = STTran(mode=mode, attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES,
               enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    m.check_indices = True
    return m


def _entries(e):
    plain = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
    guard = {k: (guarded(torch.from_numpy(v)) if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
    return plain, guard


def run_forward():
    from nl_vsgg_amd.lib.sttran import pack_clips
    sd = syn.make_sttran_state_dict(7)
    keys = ("attention_distribution", "spatial_distribution", "contacting_distribution")
    n = 0
    for mode in ("predcls", "sgdet"):
        m = _model(mode, sd)
        clips = []
        for seed, counts in ((1, [3, 1, 4, 2, 2]), (2, [11] * 16), (3, [0, 2, 0, 3]), (4, [35, 20, 35])):
            plain, guard = _entries(syn.make_entry(seed, counts, mode=mode))
            want = {k: m(dict(plain))[k].clone() for k in keys}
            got = m(dict(guard))
            m.sync_check()
            for k in keys:
                assert torch.equal(got[k], want[k]), (mode, counts, k)
            clips.append((plain, guard))
            n += 1
        if mode == "predcls":
            # the 64x36 clip: M = 2240 / 4410 rows are not multiples of the 128-row tiles of the 16x16x4 kernels
            plain, guard = _entries(syn.make_entry(5, [35] * 64, mode=mode))
            want = {k: m(dict(plain))[k].clone() for k in keys}
            got = m(dict(guard))
            m.sync_check()
            for k in keys:
                assert torch.equal(got[k], want[k]), (mode, "64x36", k)
            del plain, guard
            n += 1
            # the opt-in bf16x3 engine on every contraction
            m.gemm_engine = "bf16x3_all"
            want = {k: m(dict(clips[1][0]))[k].clone() for k in keys}
            got = m(dict(clips[1][1]))
            m.sync_check()
            for k in keys:
                assert torch.equal(got[k], want[k]), (mode, "bf16x3", k)
            m.gemm_engine = "fp32"
            n += 1
        # the by-pointer batch: every clip's tensors stay where they are (guarded), nothing is concatenated
        want = m(pack_clips([dict(c[0]) for c in clips], copy=True))
        got = m(pack_clips([dict(c[1]) for c in clips], copy=False))
        m.sync_check()
        for k in keys:
            assert torch.equal(got[k], want[k]), (mode, "batch", k)
        n += 1
    print("OK forward %d calls" % n)


def run_aux():
    """the other entry points: DSG-DETR forward, detector-output selection (f-2), union boxes (f-1), device evaluator (f-3)"""
    from nl_vsgg_amd.lib.dsg_detr import STTran as DSG
    from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    from nl_vsgg_amd.lib.union_boxes import union_boxes_and_masks
    keys = ("attention_distribution", "spatial_distribution", "contacting_distribution")
    n = 0
    m = DSG(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES).to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_dsg_detr_state_dict(7).items()}, strict=False)
    for seed, counts in ((333, [11] * 16), (334, [3, 0, 4, 1]), (335, [35, 20, 35])):
        plain, guard = _entries(syn.make_entry(seed, counts, mode="sgdet", im_idx_dtype=np.int64))
        want = {k: m(dict(plain))[k].clone() for k in keys}
        got = m(dict(guard))
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[k], want[k]), ("dsg", counts, k)
        n += 1
    for seed, counts, fd in ((301, [25, 0, 31, 18, 22], 2048), (304, [90, 80, 100], 2048), (303, [3], 2048),
                             (811, [1300, 40, 1100], 8)):              # the last: NMS tables in scratch, not in LDS
        step("select", counts)
        plain, guard = _entries(syn.make_detector_entry(seed, counts, feat_dim=fd, fmap_channels=5 if fd == 2048 else 2))
        want, got = sgdet_select(plain), sgdet_select(guard)
        torch.cuda.synchronize()
        for k, v in want.items():
            if torch.is_tensor(v):
                assert torch.equal(got[k], v), ("select", counts, k)
        n += 1
    ev = {}
    for tag in ("plain", "guard"):
        ev[tag] = SceneGraphEvaluator_HIP(mode="predcls", AG_object_classes=CLASSES, AG_all_predicates=[f"p{i}" for i in range(26)],
                                          AG_attention_predicates=[f"p{i}" for i in range(3)],
                                          AG_spatial_predicates=[f"p{i}" for i in range(3, 9)],
                                          AG_contacting_predicates=[f"p{i}" for i in range(9, 26)], iou_threshold=0.5)
        ev[tag].register_container()
    rng = np.random.default_rng(9)
    for seed, counts in ((500, [4, 2, 5, 3]), (501, [120, 100]), (502, [11] * 16), (503, [300, 2])):
        step("evaluator / union boxes", counts)
        e = syn.make_entry(seed, counts, geometry_only=True)
        gt = syn.make_gt_annotation(seed + 1, e)
        P = e["pair_idx"].shape[0]
        e["attention_distribution"] = rng.standard_normal((P, 3)).astype(np.float32)
        e["spatial_distribution"] = rng.random((P, 6)).astype(np.float32)
        e["contacting_distribution"] = rng.random((P, 17)).astype(np.float32)
        e["scores"] = np.ones(e["boxes"].shape[0], np.float32)
        plain, guard = _entries(e)
        ev["plain"].evaluate_scene_graph(gt, plain)
        ev["guard"].evaluate_scene_graph(gt, guard)
        bp, mp = union_boxes_and_masks(plain["boxes"], plain["pair_idx"])
        bg, mg = union_boxes_and_masks(guard["boxes"], guard["pair_idx"])
        assert torch.equal(bp, bg) and torch.equal(mp, mg)
        n += 1
    for tag in ev:
        ev[tag].flush()
    a, b = ev["plain"].result_dict, ev["guard"].result_dict
    for k in a:
        assert str(a[k]) == str(b[k]), ("eval", k)
    print("OK aux %d calls" % n)


def run_workspace():
    """forwards of both models at sizes whose row counts are NOT multiples of any tile; prints a digest of every output.
    Run with STTRAN_GUARD_WORKSPACE=1 (the library's own buffers and weights end at the end of their mappings) and
    without: the parent compares the digests."""
    import hashlib
    from nl_vsgg_amd.lib.dsg_detr import STTran as DSG
    from nl_vsgg_amd.lib.sttran import pack_clips
    keys = ("attention_distribution", "spatial_distribution", "contacting_distribution")

    class _H:
        """sha256 of everything + a short per-case line on stderr (which case differs, when the parent's comparison fails)"""
        def __init__(self):
            self.all, self.case, self.n = hashlib.sha256(), hashlib.sha256(), 0
        def update(self, b):
            self.all.update(b); self.case.update(b)
        def mark(self, what):
            print("case %d %s %s" % (self.n, what, self.case.hexdigest()[:16]), file=sys.stderr)
            self.case, self.n = hashlib.sha256(), self.n + 1
        def hexdigest(self):
            return self.all.hexdigest()
    h = _H()
    sd = syn.make_sttran_state_dict(7)
    for mode in ("predcls", "sgdet"):
        m = _model(mode, sd)
        sizes = ([35] * 64, [3, 1, 4, 2, 2], [11] * 16, [0, 2, 0, 3], [7] * 30) if mode == "predcls" else ([11] * 16, [5, 0, 9])
        clips = []
        for seed, counts in enumerate(sizes):                 # biggest first: every later call fits the workspace exactly or not at all
            e = syn.make_entry(seed + 20, counts, mode=mode)
            e = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
            out = m(dict(e))
            m.sync_check()
            for k in keys:
                h.update(out[k].cpu().numpy().tobytes())
            h.mark("%s %d frames" % (mode, len(counts)))
            if len(counts) < 64:
                clips.append(e)
        m2 = _model(mode, sd)                                  # a fresh handle: its workspace is sized by THIS call exactly
        out = m2(pack_clips([dict(c) for c in clips], copy=False))
        m2.sync_check()
        for k in keys:
            h.update(out[k].cpu().numpy().tobytes())
        h.mark(mode + " by-pointer batch")
        if mode == "predcls":
            m2.gemm_engine = "bf16x3_all"
            out = m2(dict(clips[1]))
            m2.sync_check()
            for k in keys:
                h.update(out[k].cpu().numpy().tobytes())
            h.mark("bf16x3_all")
    d = DSG(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES).to("cuda:0")
    d.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_dsg_detr_state_dict(7).items()}, strict=False)
    for seed, counts in ((333, [35] * 20), (334, [3, 0, 4, 1]), (335, [11] * 16)):
        e = syn.make_entry(seed, counts, mode="sgdet", im_idx_dtype=np.int64)
        e = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
        out = d(e)
        torch.cuda.synchronize()
        for k in keys:
            h.update(out[k].cpu().numpy().tobytes())
        h.mark("dsgdetr %d" % seed)
    print("OK workspace SUM " + h.hexdigest())


if __name__ == "__main__":
    what = sys.argv[1]
    torch.zeros(1, device="cuda")
    if what == "probe":
        g = guarded(torch.arange(1000, dtype=torch.float32))
        assert g.sum().item() == 499500.0
        print("OK probe")
    elif what == "gemm":
        run_gemm()
    elif what == "forward":
        run_forward()
    elif what == "aux":
        run_aux()
    elif what == "workspace":
        run_workspace()
