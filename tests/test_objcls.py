"""SURVEY 8f-2 -- SGDet without weak supervision (`lib/sttran.py:185-283`).

CPU: the oracle's restatement of the branch equals the fixtures the REFERENCE's own Python produced
(`tests/golden/gen_golden_objcls.py`; its two compiled ops stubbed by the oracle's NMS / ROIAlign restatements, which
stay parity-unpinned), plus known-answer checks of those two restatements.
GPU: the HIP path through the C ABI -- keep / pair / label indices bit-exact, float outputs to 1e-5."""
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn
from oracle import objcls_oracle as oc

CASES = ["basic", "empty_frame", "crowded", "single_frame"]
INT_KEYS = ("pred_labels", "pair_idx", "human_idx")
EXACT_KEYS = ("boxes", "distribution", "features", "pred_scores", "im_idx", "union_box")


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"objcls_{name}.npz"))
    e = syn.make_detector_entry(int(g["seed"]), g["boxes_per_frame"].tolist())
    return g, e


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference_python(name, golden_dir):
    g, e = _load(golden_dir, name)
    out = oc.objcls_select(e["boxes"], e["distribution"], e["features"], e["pred_labels"])
    for k in INT_KEYS + EXACT_KEYS:
        np.testing.assert_array_equal(np.asarray(out[k]), g[k], err_msg=k)
    rois = out["union_box"]
    uf = oc.roi_align(e["fmaps"], rois)
    np.testing.assert_array_equal(uf, g["union_feat"])
    assert out["boxes"].shape[0] != e["boxes"].shape[0]            # clean_class / NMS did change the box set
    if name == "empty_frame":
        assert g["human_idx"][2] == 0 and g["pred_labels"][0] == 1   # the reference's empty-frame assignment to row 0


def test_nms_known_answers():
    """hand-checkable cases of the +1-pixel IoU and of the two comparison flavours (nms.cu:58 '>' vs nms_cpu.cpp:62 '>=')"""
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 9], [100, 100, 109, 109], [0, 0, 9, 4]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    assert oc.nms(b, s, 0.6).tolist() == [0, 2, 3]                 # identical box dropped; half-height box IoU = 0.5
    assert oc.nms(b, s, 0.5).tolist() == [0, 2, 3]                 # IoU == 0.5 exactly: '>' keeps it ...
    assert oc.nms(b, s, 0.5, ge=True).tolist() == [0, 2]           # ... '>=' drops it
    assert oc.nms(b[::-1].copy(), s[::-1].copy(), 0.6).tolist() == [0, 1, 3]   # indices refer to the INPUT order
    assert oc.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.6).shape == (0,)
    # a chain: 0 suppresses 1, so 1 cannot suppress 2 (greedy, not transitive)
    c = np.array([[0, 0, 99, 99], [30, 0, 129, 99], [60, 0, 159, 99]], np.float32)
    assert oc.nms(c, np.array([3, 2, 1], np.float32), 0.5).tolist() == [0, 2]


def test_roi_align_known_answers():
    """a constant map pools to the constant; a linear ramp pools to the ramp at the bin centres (bilinear is exact on it)"""
    H, W = 20, 30
    const = np.full((1, 2, H, W), 3.5, np.float32)
    rois = np.array([[0, 32, 48, 200, 180], [0, 0, 0, 15, 15]], np.float32)
    np.testing.assert_allclose(oc.roi_align(const, rois), 3.5, rtol=1e-6)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ramp = (2.0 * xx + 3.0 * yy)[None, None]
    out = oc.roi_align(ramp.astype(np.float32), rois[:1])
    x1, y1, x2, y2 = rois[0, 1:] / 16.0
    cx = x1 + (np.arange(7) + 0.5) * (x2 - x1) / 7
    cy = y1 + (np.arange(7) + 0.5) * (y2 - y1) / 7
    np.testing.assert_allclose(out[0, 0], 2.0 * cx[None, :] + 3.0 * cy[:, None], rtol=1e-5)
    # a roi outside the map reads zeros; a degenerate roi is forced to 1 x 1
    far = np.array([[0, 5000, 5000, 5100, 5100]], np.float32)
    assert np.all(oc.roi_align(const, far) == 0)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _to_cuda(e):
    import torch
    return {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in e.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_select_equals_reference_fixture(name, golden_dir):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    g, e = _load(golden_dir, name)
    out = sgdet_select(_to_cuda(e))
    torch.cuda.synchronize()
    for k in INT_KEYS:
        np.testing.assert_array_equal(out[k].cpu().numpy().reshape(g[k].shape), g[k], err_msg=k)
    for k in EXACT_KEYS:                                             # copies / comparisons of float32 inputs: exact
        np.testing.assert_array_equal(out[k].cpu().numpy(), g[k], err_msg=k)
    np.testing.assert_allclose(out["union_feat"].cpu().numpy(), g["union_feat"], rtol=1e-5, atol=1e-6)
    rois = np.concatenate([g["boxes"][g["pair_idx"][:, 0], 1:], g["boxes"][g["pair_idx"][:, 1], 1:]], axis=1)
    np.testing.assert_allclose(out["spatial_masks"].cpu().numpy(), syn.union_box_masks(rois, 27) - np.float32(0.5), atol=1e-6)
    src = out["_source_row"].cpu().numpy()
    np.testing.assert_array_equal(e["boxes"][src, 1:], g["boxes"][:, 1:])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,counts,ge", [(301, [25, 0, 31, 18, 22], False), (302, [60, 55], True), (303, [3], False),
                                            (304, [90, 80, 100], False)])
def test_hip_select_equals_oracle_fresh_seeds(seed, counts, ge):
    """fresh detector outputs (incl. a frame without boxes, the >= flavour, ~100 boxes per frame at 2048-d features)"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    e = syn.make_detector_entry(seed, counts, feat_dim=2048, fmap_channels=5)
    ref = oc.objcls_select(e["boxes"], e["distribution"], e["features"], e["pred_labels"], ge=ge)
    out = sgdet_select(_to_cuda(e), nms_ge=ge)
    torch.cuda.synchronize()
    for k in INT_KEYS + EXACT_KEYS:
        np.testing.assert_array_equal(out[k].cpu().numpy().reshape(np.asarray(ref[k]).shape), ref[k], err_msg=k)
    uf = oc.roi_align(e["fmaps"], ref["union_box"][:40])
    np.testing.assert_allclose(out["union_feat"][:40].cpu().numpy(), uf, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_select_equals_oracle_many_seeds():
    """40 random detector outputs (1..6 frames, 0..40 boxes per frame, both NMS comparison flavours): selection, order,
    labels, humans and pairs bit-identical to the oracle (small features: only the index work is under test)"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    rng = np.random.default_rng(77)
    for trial in range(40):
        counts = [int(c) for c in rng.integers(0, 41, int(rng.integers(1, 7)))]
        if counts[-1] == 0:
            counts[-1] = int(rng.integers(1, 41))      # the reference sizes the clip by its LAST box (lib/sttran.py:196)
        ge = bool(trial & 1)
        e = syn.make_detector_entry(900 + trial, counts, feat_dim=8, fmap_channels=2)
        ref = oc.objcls_select(e["boxes"], e["distribution"], e["features"], e["pred_labels"], ge=ge)
        out = sgdet_select(_to_cuda(e), nms_ge=ge)
        torch.cuda.synchronize()
        for k in INT_KEYS + EXACT_KEYS:
            np.testing.assert_array_equal(out[k].cpu().numpy().reshape(np.asarray(ref[k]).shape), ref[k],
                                          err_msg=f"trial {trial} counts {counts} ge {ge}: {k}")


@pytest.mark.gpu
def test_hip_roi_align_full_width():
    """2048 channels at the detector's feature-map size, rois touching and crossing the borders"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.object_classifier import roi_align
    rng = np.random.default_rng(5)
    fm = rng.standard_normal((2, 2048, 38, 50)).astype(np.float32)
    rois = np.array([[0, 10.5, 20.25, 300.75, 400.5], [1, -20, -20, 40, 40], [1, 700, 500, 900, 700], [0, 100, 100, 101, 101],
                     [1, 0, 0, 799, 599]], np.float32)
    got = roi_align(torch.from_numpy(fm).cuda(), torch.from_numpy(rois).cuda()).cpu().numpy()
    ref = oc.roi_align(fm[:, :64], rois)                             # the oracle is slow: 64 of the channels
    np.testing.assert_allclose(got[:, :64], ref, rtol=1e-5, atol=1e-6)
    assert np.isfinite(got).all()


@pytest.mark.gpu
def test_sttran_sgdet_without_wks_end_to_end():
    """`STTran(mode='sgdet', is_wks=False)`: select boxes / pairs on the device, then the relation transformer on the
    result -- equal to running the oracle's selection and feeding ITS entry to a predcls model by hand"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.sttran import STTran
    classes = ["__background__"] + [f"c{i}" for i in range(36)]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}
    kw = dict(attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=classes, enc_layer_num=1,
              dec_layer_num=3, transformer_mode="wk", feat_dim=2048)
    e = syn.make_detector_entry(401, [9, 12, 7], feat_dim=2048, fmap_channels=2048, fmap_hw=(12, 16), image_wh=(256.0, 192.0))
    m = STTran(mode="sgdet", is_wks=False, **kw).to("cuda:0")
    m.load_state_dict(sd, strict=False)
    out = m(_to_cuda(e))
    ref = oc.objcls_select(e["boxes"], e["distribution"], e["features"], e["pred_labels"])
    np.testing.assert_array_equal(out["pair_idx"].cpu().numpy(), ref["pair_idx"])
    np.testing.assert_array_equal(out["pred_labels"].cpu().numpy(), ref["pred_labels"])
    p = STTran(mode="predcls", is_wks=True, **kw).to("cuda:0")
    p.load_state_dict(sd, strict=False)
    hand = {"features": out["features"], "pair_idx": out["pair_idx"], "labels": out["pred_labels"], "im_idx": out["im_idx"],
            "union_feat": out["union_feat"], "spatial_masks": out["spatial_masks"]}
    want = p(hand)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), want[k].cpu().numpy())
        assert out[k].shape[0] == ref["pair_idx"].shape[0]


@pytest.mark.gpu
def test_hip_select_frames_beyond_the_lds_tables():
    """a frame with more than 1 024 expanded boxes (the NMS kernel's LDS tables): rounds 1-2 refused it
    (STTRAN_ERR_LIMIT), now its tables live in the caller's scratch -- same kernel code, bit-identical to the oracle"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    e = syn.make_detector_entry(811, [1300, 40, 1100], feat_dim=8, fmap_channels=2)
    ref = oc.objcls_select(e["boxes"], e["distribution"], e["features"], e["pred_labels"])
    out = sgdet_select(_to_cuda(e))
    torch.cuda.synchronize()
    for k in INT_KEYS + EXACT_KEYS:
        np.testing.assert_array_equal(out[k].cpu().numpy().reshape(np.asarray(ref[k]).shape), ref[k], err_msg=k)


@pytest.mark.gpu
def test_hip_select_takes_boxes_in_any_order():
    """the reference selects a frame's rows with `boxes[:, 0] == i` (lib/sttran.py:59-62,205-207): any row order, the
    order inside a frame kept.  The kernels need rows grouped by frame (STTRAN_ERR_ORDER otherwise); the wrapper then
    stable-sorts by frame id and the result equals the sorted input's"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd import _native as nat
    from nl_vsgg_amd.lib.object_classifier import sgdet_select
    e = syn.make_detector_entry(812, [12, 9, 15, 7], feat_dim=16, fmap_channels=3)
    want = sgdet_select(_to_cuda(e))
    rng = np.random.default_rng(3)
    # interleave the frames, keeping the order of the rows inside each frame
    frame = e["boxes"][:, 0].astype(int)
    by_frame = {f: list(np.nonzero(frame == f)[0]) for f in np.unique(frame)}
    seq = rng.permutation(frame)                                         # which frame every output slot takes a row from
    order = np.array([by_frame[f].pop(0) for f in seq])
    assert not np.all(np.diff(e["boxes"][order, 0]) >= 0)
    shuf = dict(e)
    for k in ("boxes", "distribution", "features", "pred_labels"):
        shuf[k] = e[k][order]
    got = sgdet_select(_to_cuda(shuf))
    torch.cuda.synchronize()
    for k in INT_KEYS + EXACT_KEYS:
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k].cpu().numpy(), err_msg=k)
    # the C ABI itself reports the order instead of silently mis-assigning rows
    lib = nat.load()
    d = _to_cuda(shuf)
    B, T = len(order), 4
    cap = 4 * B
    f32, i64 = torch.float32, torch.int64
    o = [torch.empty((cap, 5), dtype=f32, device="cuda"), torch.empty((cap, 36), dtype=f32, device="cuda"),
         torch.empty((cap,), dtype=f32, device="cuda"), torch.empty((cap,), dtype=i64, device="cuda"),
         torch.empty((cap, 2), dtype=i64, device="cuda"), torch.empty((cap,), dtype=f32, device="cuda"),
         torch.zeros((T,), dtype=i64, device="cuda")]
    nscr = int(lib.sttran_objcls_scratch_bytes(B, T))
    scratch = torch.empty((nscr,), dtype=torch.uint8, device="cuda")
    a = nat.SttranObjclsSelect(struct_size=C.sizeof(nat.SttranObjclsSelect), num_frames=T, num_boxes=B, num_cols=36, feat_dim=0,
                               nms_threshold=0.6, nms_ge=0, capacity=cap, scratch_bytes=nscr)
    a.boxes, a.distribution, a.pred_labels = d["boxes"].data_ptr(), d["distribution"].data_ptr(), d["pred_labels"].data_ptr()
    a.out_boxes, a.out_distribution, a.out_pred_scores, a.out_pred_labels = (t.data_ptr() for t in o[:4])
    a.out_pair_idx, a.out_im_idx, a.out_human_idx, a.scratch = o[4].data_ptr(), o[5].data_ptr(), o[6].data_ptr(), scratch.data_ptr()
    nb, npair = C.c_int64(0), C.c_int64(0)
    assert lib.sttran_objcls_select(C.byref(a), C.byref(nb), C.byref(npair), None) == 5      # STTRAN_ERR_ORDER
    a.num_frames = 3                                                     # sorted rows, but frame id 3 >= num_frames
    keep = _to_cuda(e)
    a.boxes, a.distribution, a.pred_labels = keep["boxes"].data_ptr(), keep["distribution"].data_ptr(), keep["pred_labels"].data_ptr()
    assert lib.sttran_objcls_select(C.byref(a), C.byref(nb), C.byref(npair), None) == 5


@pytest.mark.gpu
def test_sgdet_without_wks_through_both_evaluators():
    """f-2 output at ~100+ boxes per frame (more pairs per frame than one pass of the device evaluator's key buffer
    holds) -> relation transformer -> host AND device evaluator: identical result_dict.  Rounds 1-2: the model accepted
    such a clip and the device evaluator then refused it."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator
    from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
    from nl_vsgg_amd.lib.sttran import STTran
    classes = ["__background__"] + [f"c{i}" for i in range(36)]
    att, spa, con = [f"a{i}" for i in range(3)], [f"s{i}" for i in range(6)], [f"c{i}" for i in range(17)]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}
    m = STTran(mode="sgdet", is_wks=False, attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=classes,
               enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", feat_dim=2048).to("cuda:0")
    m.load_state_dict(sd, strict=False)
    e = syn.make_detector_entry(402, [150, 130, 160], feat_dim=2048, fmap_channels=2048, fmap_hw=(12, 16), image_wh=(256.0, 192.0))
    pred = m(_to_cuda(e))
    counts = np.bincount(pred["im_idx"].cpu().numpy().astype(int), minlength=3)
    assert counts.max() > 96, counts                                    # beyond one pass of the device evaluator
    # ground truth in the AG_Test schema on the SELECTED boxes: the human of each frame + some of its objects
    boxes = pred["boxes"].cpu().numpy(); labels = pred["pred_labels"].cpu().numpy(); pair = pred["pair_idx"].cpu().numpy()
    rng = np.random.default_rng(9)
    gt = []
    for f in range(3):
        pf = pair[pred["im_idx"].cpu().numpy().astype(int) == f]
        human = int(pf[0, 0])
        frame = [{"person_bbox": boxes[human, 1:][None, :].copy()}]
        for o in rng.choice(pf[:, 1], size=min(6, len(pf)), replace=False):
            frame.append({"class": int(labels[o]), "bbox": boxes[int(o), 1:].copy(),
                          "attention_relationship": torch.tensor([int(rng.integers(0, 3))]),
                          "spatial_relationship": torch.tensor(sorted(set(rng.integers(0, 6, 2).tolist()))),
                          "contacting_relationship": torch.tensor(sorted(set(rng.integers(0, 17, 2).tolist())))})
        gt.append(frame)
    kw = dict(mode="sgdet", AG_object_classes=classes, AG_all_predicates=att + spa + con, AG_attention_predicates=att,
              AG_spatial_predicates=spa, AG_contacting_predicates=con, iou_threshold=0.5)
    host, dev = SceneGraphEvaluator(**kw), SceneGraphEvaluator_HIP(**kw)
    host.register_container(); dev.register_container()
    host.tie_break = "index"
    host.evaluate_scene_graph(gt, pred)
    dev.evaluate_scene_graph(gt, pred)
    host.calculate_mean_recall(); dev.calculate_mean_recall()
    for t in ("recall", "recall_nogc", "semi_recall"):
        for k in (10, 20, 50):
            assert host.result_dict[f"sgdet_{t}"][k] == dev.result_dict[f"sgdet_{t}"][k], (t, k)
    for t in ("mean_recall", "ng_mean_recall"):
        for k in (10, 20, 50):
            assert host.result_dict[f"sgdet_{t}_collect"][k] == dev.result_dict[f"sgdet_{t}_collect"][k], (t, k)
    assert any(v > 0 for v in host.result_dict["sgdet_recall"][50])    # the ground truth is found at all
