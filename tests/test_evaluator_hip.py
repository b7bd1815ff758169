"""Device evaluator: host-side packing / tallying on the CPU, the matching kernel on the GPU.

The checker is the host evaluator of `lib/evaluation_recall.py`, itself pinned to recall values captured
from the reference (`tests/golden/eval_*.json`); equality is required, not a tolerance."""
import json
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn
from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator

OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]
SPA = [f"spa{i}" for i in range(6)]
CON = [f"con{i}" for i in range(17)]
KW = dict(AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON, AG_attention_predicates=ATT,
          AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)


def _clip(seed, counts, mode="predcls", jitter=0.0, pred_seed=0):
    e = syn.make_entry(seed, counts, mode=mode, im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    gt = syn.make_gt_annotation(seed + 1, e)
    rng = np.random.default_rng(pred_seed)
    if jitter:                                  # move the ground-truth boxes so that the IoU test decides
        for frame in gt:
            frame[0]["person_bbox"] = frame[0]["person_bbox"] + rng.uniform(-jitter, jitter, (1, 4)).astype(np.float32)
            for obj in frame[1:]:
                obj["bbox"] = obj["bbox"] + rng.uniform(-jitter, jitter, 4).astype(np.float32)
    P = sum(counts)
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    pred["attention_distribution"] = (3 * rng.standard_normal((P, 3))).astype(np.float32)
    pred["spatial_distribution"] = rng.random((P, 6)).astype(np.float32)
    pred["contacting_distribution"] = rng.random((P, 17)).astype(np.float32)
    pred["scores"] = rng.uniform(0.3, 1.0, len(e["labels"])).astype(np.float32)
    pred["pred_labels"], pred["pred_scores"] = pred["labels"], pred["scores"]
    return gt, pred


def _same_results(a, b, mode):
    for t in ("recall", "recall_nogc", "semi_recall"):
        for k in (10, 20, 50):
            assert a[f"{mode}_{t}"][k] == b[f"{mode}_{t}"][k], (t, k)
    for t in ("mean_recall", "ng_mean_recall"):
        for k in (10, 20, 50):
            assert a[f"{mode}_{t}"][k] == b[f"{mode}_{t}"][k], (t, k)
            assert a[f"{mode}_{t}_list"][k] == b[f"{mode}_{t}_list"][k], (t, k)
            assert a[f"{mode}_{t}_collect"][k] == b[f"{mode}_{t}_collect"][k], (t, k)


# ---- CPU: packing + tallying reproduce the host evaluator's containers from its own hit table --------
@pytest.mark.parametrize("counts", [[11] * 16, [3, 1, 4, 2, 2], [1], [6, 2]])
def test_pack_and_tally_match_host_evaluator(counts):
    pytest.importorskip("torch")
    from nl_vsgg_amd.lib.evaluation_recall_hip import pack_ground_truth, tally_hit_flags
    host = SceneGraphEvaluator(mode="predcls", **KW); host.register_container(); host.hit_flags = []
    mine = SceneGraphEvaluator(mode="predcls", **KW); mine.register_container()
    for c in range(2):
        gt, pred = _clip(40 + c, counts, jitter=25.0, pred_seed=c)
        host.evaluate_scene_graph(gt, pred)
        packed = pack_ground_truth(gt, mine)
        assert packed.num_frames == len(gt) and packed.rel_off[-1] == packed.rels.shape[0]
        flags = np.concatenate(host.hit_flags); host.hit_flags = []
        assert flags.shape == (packed.rels.shape[0], 9)
        tally_hit_flags(mine, mine.result_dict, packed, flags)
    host.calculate_mean_recall(); mine.calculate_mean_recall()
    _same_results(host.result_dict, mine.result_dict, "predcls")


def test_packed_ground_truth_round_trip():
    pytest.importorskip("torch")
    from nl_vsgg_amd.lib.evaluation_recall_hip import pack_ground_truth
    ev = SceneGraphEvaluator(mode="predcls", **KW)
    gt, _ = _clip(9, [3, 1, 4], jitter=5.0)
    p1 = pack_ground_truth(gt, ev)
    p2 = pack_ground_truth(p1.to_annotation(ev), ev)
    for k in ("box_off", "boxes", "classes", "rel_off"):
        assert np.array_equal(getattr(p1, k), getattr(p2, k)), k
    assert sorted(map(tuple, p1.rels.tolist())) == sorted(map(tuple, p2.rels.tolist()))


# ---- GPU: the kernel's hit table equals the host evaluator's ---------------------------------------------
def _device_eval(mode="predcls"):
    from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
    ev = SceneGraphEvaluator_HIP(mode=mode, **KW)
    ev.register_container()
    return ev


def _to_dev(pred):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in pred.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uniform_16x12", "ragged_5", "sgdet_ragged", "hard_empty_frames", "hard_sgdet_empty_frames",
                                  "hard_uniform_16x12", "hard_sgdet_16x12"])
def test_device_recall_identical_to_reference(case, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, f"eval_{case}.json")))
    g = np.load(os.path.join(golden_dir, f"sttran_{ref.get('fixture', case)}.npz"))
    mode = ref["mode"]
    e = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                       im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    if ref.get("gt") == "hard":      # the reference evaluator on GT it has to work for: frames without predictions, IoU near 0.5
        gt = syn.make_gt_annotation_hard(ref["gt_seed"], e, jitter=ref["jitter"])
    else:
        gt = syn.make_gt_annotation(ref["gt_seed"], e)
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        pred[k] = g[k]
    pred["pred_labels"], pred["pred_scores"] = pred["labels"], pred["scores"]
    ev = _device_eval(mode)
    ev.evaluate_scene_graph(gt, _to_dev(pred))
    ev.calculate_mean_recall()
    rd, want = ev.result_dict, ref["result_dict"]
    for t in ("recall", "recall_nogc", "semi_recall"):
        for k in (10, 20, 50):
            assert rd[f"{mode}_{t}"][k] == want[f"{mode}_{t}"][str(k)], (t, k)
    for t in ("mean_recall", "ng_mean_recall"):
        for k in (10, 20, 50):
            assert rd[f"{mode}_{t}"][k] == pytest.approx(want[f"{mode}_{t}"][str(k)], abs=1e-12)
            np.testing.assert_allclose(rd[f"{mode}_{t}_list"][k], want[f"{mode}_{t}_list"][str(k)], atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("counts,jitter", [([11] * 16, 0.0), ([11] * 16, 25.0), ([3, 1, 4, 2, 2], 25.0), ([1], 10.0),
                                           ([1, 2, 1], 0.0), ([35] * 8, 30.0), ([96, 5], 20.0),
                                           # more pairs than one pass of the key buffer holds (96 at 26 predicates): chunked
                                           ([97, 3], 20.0), ([300, 150, 7], 25.0), ([193, 192], 0.0), ([500], 15.0)])
def test_device_hit_table_equals_host(counts, jitter):
    host = SceneGraphEvaluator(mode="predcls", **KW); host.register_container(); host.tie_break = "index"
    dev = _device_eval()
    for c in range(3):                           # several clips pending before one flush
        gt, pred = _clip(70 + c, counts, jitter=jitter, pred_seed=10 + c)
        host.evaluate_scene_graph(gt, pred)
        dev.evaluate_scene_graph(dev.pack(gt), _to_dev(pred))
    host.calculate_mean_recall(); dev.calculate_mean_recall()
    _same_results(host.result_dict, dev.result_dict, "predcls")
    assert host.summary() == dev.summary()


@pytest.mark.gpu
def test_device_eval_follows_model_output():
    """model -> evaluator without leaving the device: same recall as the host evaluator on the copied-back dict"""
    import torch
    from nl_vsgg_amd.lib.sttran import STTran
    model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=OBJ,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda")
    model.eval()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()},
                          strict=False)
    host = SceneGraphEvaluator(mode="predcls", **KW); host.register_container(); host.tie_break = "index"
    dev = _device_eval()
    for c, counts in enumerate(([4, 2, 5, 3], [6] * 5)):
        e = syn.make_entry(300 + c, counts)
        gt = syn.make_gt_annotation(400 + c, e)
        entry = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v)
                 for k, v in e.items()}
        pred = model(entry)
        dev.evaluate_scene_graph(gt, pred)
        host.evaluate_scene_graph(gt, pred)
    host.calculate_mean_recall(); dev.calculate_mean_recall()
    _same_results(host.result_dict, dev.result_dict, "predcls")


@pytest.mark.gpu
def test_device_eval_has_no_pairs_per_frame_limit():
    """rounds 1-2 refused frames with more than 96 pairs (one pass of the 58.5 KB key buffer); a frame now goes through
    the buffer in chunks whose top-50 lists are merged -- 300 pairs per frame, all three metrics, equal to the host"""
    dev = _device_eval()
    assert dev.max_pairs_per_frame == 96                         # pairs per PASS, no longer a limit
    host = SceneGraphEvaluator(mode="predcls", **KW); host.register_container(); host.tie_break = "index"
    gt, pred = _clip(5, [300, 300, 97, 96, 1], jitter=20.0, pred_seed=3)
    dev.evaluate_scene_graph(gt, _to_dev(pred))
    host.evaluate_scene_graph(gt, pred)
    host.calculate_mean_recall(); dev.calculate_mean_recall()     # flush: no error
    _same_results(host.result_dict, dev.result_dict, "predcls")


@pytest.mark.gpu
def test_device_eval_needs_device_predictions():
    dev = _device_eval()
    gt, pred = _clip(5, [2, 2])
    with pytest.raises(RuntimeError):
        dev.evaluate_scene_graph(gt, pred)
