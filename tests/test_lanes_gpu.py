"""Lanes (include/sttran_hip.h "LANES"): several forwards of ONE handle in flight on the handle's own streams -- how the
reference's one-clip-per-call loop (tools/test_STTran.py:81-92) keeps the MI355X busy.  Every lane result must equal the
classic single-stream forward bit for bit; ordering is by events only (no host synchronisation inside the loop)."""
import collections
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

OUT_KEYS = ("attention_distribution", "spatial_distribution", "contacting_distribution")
CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]


def _model(mode, sd, model="sttran"):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if model == "dsgdetr":
        from nl_vsgg_amd.lib.dsg_detr import STTran
        m = STTran(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES).to("cuda:0")
    else:
        from nl_vsgg_amd.lib.sttran import STTran
        m = STTran(mode=mode, attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    m.check_indices = False
    return m


def _cuda_entry(e):
    return {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}


SHAPES = [[3, 1, 4, 2, 2], [11] * 16, [2, 0, 3, 0, 0, 2], [5], [1, 1, 1], [7, 9, 8, 2, 6, 6, 6, 1, 4], [35] * 6, [4, 4]]


@pytest.fixture(scope="module")
def weights():
    return syn.make_sttran_state_dict(7)


@pytest.mark.parametrize("mode", ["predcls", "sgdet"])
@pytest.mark.parametrize("lanes", [2, 4])
def test_lane_forwards_equal_the_classic_forward(weights, mode, lanes):
    m = _model(mode, weights)
    entries = [syn.make_entry(300 + i, SHAPES[i % len(SHAPES)], mode=mode) for i in range(3 * len(SHAPES))]
    want = []
    for e in entries:
        p = m(_cuda_entry(e))
        want.append({k: p[k].clone() for k in OUT_KEYS + (("distribution",) if mode == "sgdet" else ())})
    torch.cuda.synchronize()
    m.lanes = lanes
    pending, got = collections.deque(), []
    for e in entries:                                     # the pipelined form of the reference's loop: no synchronisation
        pending.append(m.forward_async(_cuda_entry(e)))
        if len(pending) == lanes:
            p = m.join(pending.popleft())
            got.append({k: p[k].clone() for k in want[0]})  # a consumer on the current stream, right behind the join
    while pending:
        p = m.join(pending.popleft())
        got.append({k: p[k].clone() for k in want[0]})
    m.sync_check()
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        for k in w:
            assert torch.equal(g[k], w[k]), (i, k)
    # lanes were really used round robin, and the classic call still works on a multi-lane handle
    assert [int(x) for x in {i % lanes for i in range(len(entries))}] == list(range(lanes))
    p = m(_cuda_entry(entries[1]))
    torch.cuda.synchronize()
    assert all(torch.equal(p[k], want[1][k]) for k in OUT_KEYS)
    m.lanes = 1                                           # back to one lane: the extra workspaces are released
    p = m(_cuda_entry(entries[2]))
    torch.cuda.synchronize()
    assert all(torch.equal(p[k], want[2][k]) for k in OUT_KEYS)


def test_lane_inputs_produced_on_the_callers_stream_are_seen(weights):
    """fork semantics: a lane call sees everything enqueued on the caller's stream before it (here: the entry's tensors
    are WRITTEN on the current stream right before the call, with no synchronisation in between)"""
    m = _model("predcls", weights)
    e = syn.make_entry(11, [11] * 16)
    ce = _cuda_entry(e)
    want = {k: v.clone() for k, v in m(dict(ce)).items() if k in OUT_KEYS}
    torch.cuda.synchronize()
    m.lanes = 3
    big = torch.empty(64, 1024, 1024, device="cuda")
    for rep in range(6):
        staged = {k: (torch.empty_like(v) if isinstance(v, torch.Tensor) else v) for k, v in ce.items()}
        big.normal_()                                     # keeps the current stream busy: the copies below are queued behind it
        for k, v in ce.items():
            if isinstance(v, torch.Tensor):
                staged[k].copy_(v, non_blocking=True)
        p = m.forward_async(staged)
        m.join(p)
        out = {k: p[k].clone() for k in OUT_KEYS}
        torch.cuda.synchronize()
        assert all(torch.equal(out[k], want[k]) for k in OUT_KEYS), rep


def test_classic_forward_on_changing_streams_is_ordered_by_the_library(weights):
    """the cached index-map / chunk-table uploads belong to the stream of the call that made them; a forward
    of the same handle on ANOTHER stream is ordered behind the previous one by the library itself"""
    m = _model("predcls", weights)
    e = _cuda_entry(syn.make_entry(12, [11] * 16))
    want = {k: v.clone() for k, v in m(dict(e)).items() if k in OUT_KEYS}
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    e2 = _cuda_entry(syn.make_entry(13, [3, 1, 4, 2, 2]))
    outs = []
    for i in range(12):
        with torch.cuda.stream(streams[i % 3]):
            m(dict(e2) if i % 4 == 3 else dict(e))        # layouts alternate: the caches are refilled on changing streams
            p = m(dict(e))
            outs.append({k: p[k] for k in OUT_KEYS})
    torch.cuda.synchronize()
    for o in outs:
        assert all(torch.equal(o[k], want[k]) for k in OUT_KEYS)


def test_lanes_by_pointer_batches_and_dsg_detr():
    """the packed (by pointer) form and the second model on lanes"""
    from nl_vsgg_amd.lib.sttran import pack_clips
    sd = syn.make_dsg_detr_state_dict(7)
    m = _model("sgdet", sd, model="dsgdetr")
    packs = []
    for j in range(5):
        clips = [_cuda_entry(syn.make_entry(500 + 10 * j + i, SHAPES[(i + j) % len(SHAPES)], mode="sgdet", im_idx_dtype=np.int64))
                 for i in range(3)]
        packs.append(clips)
    want = []
    for clips in packs:
        p = m(pack_clips([dict(c) for c in clips], copy=False))
        want.append({k: p[k].clone() for k in OUT_KEYS})
    torch.cuda.synchronize()
    m.lanes = 2
    preds = [m.forward_async(pack_clips([dict(c) for c in clips], copy=False)) for clips in packs]
    m.sync_check()                                        # joins every lane
    for p, w in zip(preds, want):
        assert all(torch.equal(p[k], w[k]) for k in OUT_KEYS)


def test_lane_argument_errors(weights):
    import ctypes as C
    from nl_vsgg_amd import _native as nat
    m = _model("predcls", weights)
    m(_cuda_entry(syn.make_entry(1, [2, 2])))
    lib, h = m._lib, m._handle
    assert lib.sttran_num_lanes(h) == 1
    assert lib.sttran_set_lanes(h, 0) == nat.STTRAN_ERR_INVALID and lib.sttran_set_lanes(h, 9) == nat.STTRAN_ERR_INVALID
    assert lib.sttran_lane_join(h, 1, None) == nat.STTRAN_ERR_INVALID
    inp = nat.SttranInputs(struct_size=C.sizeof(nat.SttranInputs))
    out = nat.SttranOutputs(struct_size=C.sizeof(nat.SttranOutputs))
    assert lib.sttran_forward_lane(h, 3, C.byref(inp), C.byref(out), None) == nat.STTRAN_ERR_INVALID
    assert lib.sttran_set_lanes(h, 3) == 0 and lib.sttran_num_lanes(h) == 3
    p = C.c_void_p()
    assert lib.sttran_lane_stream(h, 2, C.byref(p)) == 0 and p.value
    assert lib.sttran_lane_join(h, -1, None) == 0         # nothing in flight: a no-op
    with pytest.raises(ValueError):
        m.lanes = 0


def test_first_calls_of_fresh_lanes_see_initialised_buffers(weights):
    """A lane's workspace, chunk table and index staging are allocated (and zeroed) inside its FIRST forward.  hipMemset runs
    in the NULL stream, which the lane's non-blocking stream does not follow: round 4 shipped a version whose zeroing could
    land after the chunk table's upload (an intermittent GPU fault on address nil, 1 run in ~8 of the one-clip bench leg).
    Fresh handles, no reserve(), lanes used at once -- every result must equal the classic forward's."""
    e_small = syn.make_entry(21, [3, 1, 4, 2, 2])
    e_big = syn.make_entry(22, [11] * 16)
    ref = _model("predcls", weights)
    want = {}
    for name, e in (("small", e_small), ("big", e_big)):
        p = ref(_cuda_entry(e))
        want[name] = {k: p[k].clone() for k in OUT_KEYS}
    torch.cuda.synchronize()
    for rep in range(6):
        m = _model("predcls", weights)
        m.lanes = 3
        preds = [(n, m.forward_async(_cuda_entry(e))) for n, e in (("small", e_small), ("big", e_big), ("small", e_small),
                                                                     ("big", e_big), ("big", e_big), ("small", e_small))]
        m.sync_check()
        for n, p in preds:
            assert all(torch.equal(p[k], want[n][k]) for k in OUT_KEYS), (rep, n)
        m._destroy()


@pytest.mark.parametrize("coalesce", [0, 3])
def test_second_engine_on_lanes_and_engine_switches(weights, coalesce):
    """the bf16x3 engine keeps per-lane plane buffers (activation planes written by LayerNorm / the linear1 and linear2
    epilogues, `hplanes_of` tracking): forwards in flight on several lanes, coalesced or not, equal the classic forward of
    the same engine bit for bit (packed groups: to fp32 rounding, the batch changes the tiling); switching the engine
    between calls (fp32 -> bf16x3 -> fp32 -> bf16x3) reuses the planes made at the first switch"""
    m = _model("predcls", weights)
    shapes = [[11] * 16, [35] * 6, [3, 1, 4, 2, 2], [11] * 16, [7, 9, 8, 2, 6, 6, 6, 1, 4], [35] * 6]
    entries = [syn.make_entry(800 + i, s) for i, s in enumerate(shapes)]
    want32 = [{k: m(_cuda_entry(e))[k].clone() for k in OUT_KEYS} for e in entries]
    m.gemm_engine = "bf16x3_all"
    want = [{k: m(_cuda_entry(e))[k].clone() for k in OUT_KEYS} for e in entries]
    torch.cuda.synchronize()
    for w3, w in zip(want32, want):
        assert all(float((w3[k] - w[k]).abs().max()) < 2e-5 for k in OUT_KEYS)
    m.lanes, m.coalesce = 3, coalesce
    pending, got = collections.deque(), []
    for rep in range(2):
        for e in entries:
            pending.append(m.forward_async(_cuda_entry(e)))
            if len(pending) == m.pipeline_depth:
                p = m.join(pending.popleft())
                got.append({k: p[k].clone() for k in OUT_KEYS})
    while pending:
        p = m.join(pending.popleft())
        got.append({k: p[k].clone() for k in OUT_KEYS})
    m.sync_check()
    assert len(got) == 2 * len(entries)
    for i, g in enumerate(got):
        w = want[i % len(entries)]
        for k in OUT_KEYS:
            if coalesce:
                assert float((g[k] - w[k]).abs().max()) < 2e-5, (i, k)
            else:
                assert torch.equal(g[k], w[k]), (i, k)
    m.coalesce, m.lanes = 0, 1
    for eng, ref in (("fp32", want32), ("bf16x3_all", want), ("fp32", want32), ("bf16x3", want32)):
        m.gemm_engine = eng
        p = m(_cuda_entry(entries[1]))
        torch.cuda.synchronize()
        tol = 0.0 if eng != "bf16x3" else 2e-5             # "bf16x3": >= 512 rows only -- this clip (210 pairs) mixes both engines
        assert all(float((p[k] - ref[1][k]).abs().max()) <= tol for k in OUT_KEYS), eng
