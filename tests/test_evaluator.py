"""The package's SceneGraphEvaluator against recall values captured from the reference evaluator
(tests/golden/gen_golden_eval.py).  CPU only; identical numbers are required, not a tolerance."""
import json
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn
from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator, box_iou_plus1

OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]
SPA = [f"spa{i}" for i in range(6)]
CON = [f"con{i}" for i in range(17)]


def _run(case, golden_dir, perturb=0.0):
    ref = json.load(open(os.path.join(golden_dir, f"eval_{case}.json")))
    g = np.load(os.path.join(golden_dir, f"sttran_{ref.get('fixture', case)}.npz"))
    mode = ref["mode"]
    e = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                       im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    if ref.get("gt") == "hard":      # every frame present (also those without a predicted pair), jittered boxes, flipped classes
        gt = syn.make_gt_annotation_hard(ref["gt_seed"], e, jitter=ref["jitter"])
    else:
        gt = syn.make_gt_annotation(ref["gt_seed"], e)
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        pred[k] = g[k] + np.float32(perturb)
    pred["pred_labels"], pred["pred_scores"] = pred["labels"], pred["scores"]
    ev = SceneGraphEvaluator(mode=mode, AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON,
                             AG_attention_predicates=ATT, AG_spatial_predicates=SPA, AG_contacting_predicates=CON,
                             iou_threshold=0.5)
    ev.register_container()
    ev.evaluate_scene_graph(gt, pred)
    ev.calculate_mean_recall()
    return ev, ref["result_dict"], mode


HARD = ["hard_empty_frames", "hard_sgdet_empty_frames", "hard_uniform_16x12", "hard_sgdet_16x12"]


@pytest.mark.parametrize("case", ["uniform_16x12", "ragged_5", "sgdet_ragged"] + HARD)
def test_recall_identical_to_reference(case, golden_dir):
    ev, ref, mode = _run(case, golden_dir)
    for t in ("recall", "recall_nogc", "semi_recall"):
        for k in (10, 20, 50):
            assert ev.result_dict[f"{mode}_{t}"][k] == ref[f"{mode}_{t}"][str(k)], (t, k)
    for t in ("mean_recall", "ng_mean_recall"):
        for k in (10, 20, 50):
            assert ev.result_dict[f"{mode}_{t}"][k] == pytest.approx(ref[f"{mode}_{t}"][str(k)], abs=1e-12)
            np.testing.assert_allclose(ev.result_dict[f"{mode}_{t}_list"][k], ref[f"{mode}_{t}_list"][str(k)], atol=1e-12)


def test_recall_stable_under_1e3_logit_noise(golden_dir):
    """Recall@K is what the 1e-3 logit tolerance protects: a uniform 1e-4 shift changes nothing."""
    a, _, mode = _run("uniform_16x12", golden_dir)
    b, _, _ = _run("uniform_16x12", golden_dir, perturb=1e-4)
    assert a.summary()["recall"] == b.summary()["recall"]


def test_iou_plus_one_convention():
    # identical boxes -> 1; touching boxes overlap by the +1 pixel column
    assert box_iou_plus1([0, 0, 9, 9], np.array([[0, 0, 9, 9]]))[0] == 1.0
    v = box_iou_plus1([0, 0, 9, 9], np.array([[9, 0, 18, 9], [20, 20, 30, 30]]))
    assert v[0] == pytest.approx(10.0 / 190.0) and v[1] == 0.0


def test_print_stats_runs(golden_dir):
    ev, _, _ = _run("ragged_5", golden_dir)
    text = ev.print_stats()
    assert "R @ 20" in text and "type=Recall(Main)" in text


def test_evaluate_packed_equals_per_clip_calls():
    """a `pack_clips` entry scored in one call == its clips scored one by one (host evaluator, CPU tensors)"""
    torch = pytest.importorskip("torch")
    from nl_vsgg_amd.lib.sttran import pack_clips
    rng = np.random.default_rng(3)
    clips, gts, preds = [], [], []
    for c, counts in enumerate([[3, 1, 4], [2, 2], [1, 5, 2, 2]]):
        e = syn.make_entry(640 + c, counts, geometry_only=True)
        P = sum(counts)
        e["features"] = np.zeros((e["boxes"].shape[0], 4), np.float32)          # pack_clips sizes the box offsets by it
        gts.append(syn.make_gt_annotation(1640 + c, e))
        d = {"attention_distribution": (3 * rng.standard_normal((P, 3))).astype(np.float32),
             "spatial_distribution": rng.random((P, 6)).astype(np.float32),
             "contacting_distribution": rng.random((P, 17)).astype(np.float32)}
        preds.append(d)
        clips.append({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()})
    kw = dict(mode="predcls", AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON, AG_attention_predicates=ATT,
              AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)
    one, many = SceneGraphEvaluator(**kw), SceneGraphEvaluator(**kw)
    one.register_container(); many.register_container()
    for e, gt, d in zip(clips, gts, preds):
        p = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
        p.update(d)
        one.evaluate_scene_graph(gt, p)
    packed = pack_clips(clips)
    for k in preds[0]:
        packed[k] = np.concatenate([d[k] for d in preds])
    many.evaluate_packed(gts, packed)
    one.calculate_mean_recall(); many.calculate_mean_recall()
    for key in one.result_dict:
        assert one.result_dict[key] == many.result_dict[key], key
    with pytest.raises(ValueError):
        many.evaluate_packed(gts[:2], packed)


def test_tie_break_by_index_only_differs_on_exact_ties():
    """`tie_break = "index"` (the device evaluator's documented order) gives the reference-pinned results wherever no two
    candidates of a frame score exactly alike, and a defined result -- lower candidate index first -- where they do"""
    kw = dict(mode="predcls", AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON, AG_attention_predicates=ATT,
              AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)
    e = syn.make_entry(31, [4, 3, 5])
    gt = syn.make_gt_annotation(32, e)
    rng = np.random.default_rng(1)
    P = 12
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    pred["attention_distribution"] = rng.standard_normal((P, 3)).astype(np.float32)
    pred["spatial_distribution"] = rng.random((P, 6)).astype(np.float32)
    pred["contacting_distribution"] = rng.random((P, 17)).astype(np.float32)
    a, b = SceneGraphEvaluator(**kw), SceneGraphEvaluator(**kw)
    b.tie_break = "index"
    for ev in (a, b):
        ev.register_container(); ev.evaluate_scene_graph(gt, pred); ev.calculate_mean_recall()
    assert a.result_dict == b.result_dict                              # distinct random floats: no ties, same answer
    # saturated probabilities: every spatial / contacting entry exactly 1.0 -> the order is the index order
    pred["spatial_distribution"][:] = 1.0
    pred["contacting_distribution"][:] = 1.0
    c = SceneGraphEvaluator(**kw); c.tie_break = "index"
    c.register_container(); c.evaluate_scene_graph(gt, pred)
    c2 = SceneGraphEvaluator(**kw); c2.tie_break = "index"
    c2.register_container(); c2.evaluate_scene_graph(gt, {k: (v.copy() if hasattr(v, "copy") else v) for k, v in pred.items()})
    assert c.result_dict == c2.result_dict
