"""BASELINE.json configs[0]: "STTran PredCls on 4 pre-extracted Action Genome clips, CPU reference path
(tools/test_STTran.py, no GPU)".  The fixture (tests/golden/gen_golden_ag4.py) holds what the REFERENCE model and the
REFERENCE evaluator produced on the CPU for the loop of `tools/test_STTran.py:75-92` over four AG-test-split-shaped
clips (3 / 18 / 29 / 43 frames, 0..6 pairs per frame), one clip per forward, one evaluator over all four.

CPU tests: the oracle reproduces the four outputs; this package's evaluator, fed the reference's outputs, reproduces
the reference's `result_dict`.  GPU test: the same loop on the HIP path -- outputs within the north-star 1e-3, and the
recall lists IDENTICAL (host and device evaluator)."""
import json
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn
from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator

OUT_KEYS = ("attention_distribution", "spatial_distribution", "contacting_distribution")
OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]
SPA = [f"spa{i}" for i in range(6)]
CON = [f"con{i}" for i in range(17)]
KW = dict(mode="predcls", AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON, AG_attention_predicates=ATT,
          AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "ag4_reference_loop.npz"))
    ref = json.load(open(os.path.join(golden_dir, "ag4_reference_loop.json")))
    clips = []
    for i in range(ref["clips"]):
        seed = int(g[f"clip{i}_seed"])
        e = syn.make_entry(seed, g[f"clip{i}_pairs_per_frame"].tolist())
        clips.append((e, syn.make_gt_annotation(ref["gt_seed_base"] + seed, e), {k: g[f"clip{i}_{k}"] for k in OUT_KEYS}))
    return clips, ref["result_dict"], int(g["weight_seed"])


def _same_results(ev, ref):
    for t in ("recall", "recall_nogc", "semi_recall"):
        for k in (10, 20, 50):
            assert ev.result_dict[f"predcls_{t}"][k] == ref[f"predcls_{t}"][str(k)], (t, k)
    for t in ("mean_recall", "ng_mean_recall"):
        for k in (10, 20, 50):
            assert ev.result_dict[f"predcls_{t}"][k] == pytest.approx(ref[f"predcls_{t}"][str(k)], abs=1e-12)


def _pred_of(e, dists):
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    pred.update(dists)
    pred["pred_labels"], pred["pred_scores"] = pred["labels"], pred["scores"]
    return pred


def test_shape_is_the_split_s(golden_dir):
    clips, _, _ = _load(golden_dir)
    lengths = sorted(json.load(open(os.path.join(golden_dir, "ag_test_clip_lengths.json")))["frames_per_clip"])
    frames = [int(e["num_frames"]) for e, _, _ in clips]
    assert frames == [lengths[0], lengths[len(lengths) // 2], lengths[len(lengths) // 4], lengths[3 * len(lengths) // 4]]
    assert any(0 in e["frame_counts"] for e, _, _ in clips)            # frames without a pair inside the clips


def test_evaluator_reproduces_reference_loop(golden_dir):
    clips, ref, _ = _load(golden_dir)
    ev = SceneGraphEvaluator(**KW)
    ev.register_container()
    for e, gt, dists in clips:
        ev.evaluate_scene_graph(gt, _pred_of(e, dists))
    ev.calculate_mean_recall()
    _same_results(ev, ref)
    assert len(ev.result_dict["predcls_recall"][20]) == sum(len(gt) for _, gt, _ in clips)


def test_oracle_reproduces_reference_loop(golden_dir):
    from oracle import sttran_oracle as orc
    clips, ref, wseed = _load(golden_dir)
    sd = syn.make_sttran_state_dict(wseed)
    ev = SceneGraphEvaluator(**KW)
    ev.register_container()
    for e, gt, dists in clips:
        out = orc.sttran_forward(e, sd)
        for k in OUT_KEYS:
            np.testing.assert_allclose(out[k], dists[k], atol=2e-5, rtol=0, err_msg=k)
        ev.evaluate_scene_graph(gt, _pred_of(e, {k: out[k] for k in OUT_KEYS}))
    ev.calculate_mean_recall()
    _same_results(ev, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["fp32", "bf16x3_all"])
def test_hip_reproduces_reference_loop(golden_dir, engine):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
    from nl_vsgg_amd.lib.sttran import STTran
    clips, ref, wseed = _load(golden_dir)
    sd = syn.make_sttran_state_dict(wseed)
    model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=OBJ,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    model.eval()
    model.gemm_engine = engine            # exact fp32 MFMA, and the opt-in bf16x3 emulation: same checks, same tolerances
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    ev_host, ev_dev = SceneGraphEvaluator(**KW), SceneGraphEvaluator_HIP(**KW)
    ev_host.register_container(); ev_dev.register_container()
    for e, gt, dists in clips:                                         # tools/test_STTran.py:75-88, one clip per forward
        entry = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v)
                 for k, v in e.items()}
        pred = model(entry)
        torch.cuda.synchronize()
        for k in OUT_KEYS:
            np.testing.assert_allclose(pred[k].cpu().numpy(), dists[k], atol=1e-3, rtol=0, err_msg=k)
        ev_host.evaluate_scene_graph(gt, pred)
        ev_dev.evaluate_scene_graph(gt, pred)
    ev_host.calculate_mean_recall(); ev_dev.calculate_mean_recall()
    _same_results(ev_host, ref)
    _same_results(ev_dev, ref)
