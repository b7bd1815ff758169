"""BASELINE.json configs[2] -- the loop of `tools/test_STTran.py:75-92` over the Action Genome test split, PredCls --
on a sample of AG-test-split-shaped synthetic clips (the real annotations / features / checkpoint are not shipped
with the reference): frames per clip from `datasets/AG/ag_test_id.pkl`
(tests/golden/ag_test_clip_lengths.json), 1..6 pairs per frame, `spatial_masks` from the boxes by the f-1 kernel,
clips packed 16 per forward, predictions straight into the device evaluator.

Checked: (1) the device evaluator's `result_dict` equals the host evaluator's on the same predictions, list by
list -- called once per clip and once per pack of 16 clips (`evaluate_packed`); (2) the HIP outputs of sampled clips (the 121-frame one, a 3-frame one and two others) are within 1e-3 of the
fp64 oracle on the same inputs; (3) a clip's packed result equals its single-clip result to rounding."""
import json
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

OUT_KEYS = ("attention_distribution", "spatial_distribution", "contacting_distribution")
N_CLIPS, PACK = 64, 16


def _entry_to_numpy(e):
    out = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in e.items()}
    return out


@pytest.mark.parametrize("engine", ["fp32", "bf16x3"])
def test_ag_split_shaped_loop(golden_dir, engine):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ag_split_bench as ag
    from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator
    from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
    from nl_vsgg_amd.lib.sttran import STTran, pack_clips, unpack_predictions
    from oracle import sttran_oracle as orc

    lengths = json.load(open(os.path.join(golden_dir, "ag_test_clip_lengths.json")))["frames_per_clip"]
    # a deterministic sample of the split that holds its extremes: the 121-frame clip, a 3-frame clip
    i_long, i_short = lengths.index(max(lengths)), lengths.index(min(lengths))
    assert lengths[i_long] == 121 and lengths[i_short] == 3
    rng_pick = np.random.default_rng(7)
    rest = [int(i) for i in rng_pick.permutation(len(lengths)) if i not in (i_long, i_short)][: N_CLIPS - 2]
    picked = [i_long, i_short] + rest
    picked.sort(key=lambda i: -lengths[i])                      # longest first, like the tool

    dev = torch.device("cuda", 0)
    sd = syn.make_sttran_state_dict(7)
    model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=ag.OBJ,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
    model.eval()
    model.gemm_engine = engine            # exact fp32 MFMA, and the opt-in bf16x3 emulation: same checks, same tolerances
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    kw = dict(mode="predcls", AG_object_classes=ag.OBJ, AG_all_predicates=ag.ATT + ag.SPA + ag.CON,
              AG_attention_predicates=ag.ATT, AG_spatial_predicates=ag.SPA, AG_contacting_predicates=ag.CON, iou_threshold=0.5)
    ev_dev, ev_host = SceneGraphEvaluator_HIP(**kw), SceneGraphEvaluator(**kw)
    ev_host.tie_break = "index"       # exactly equal scores: the device's documented order (numpy's is an accident of its sort)
    ev_dev.register_container(); ev_host.register_container()

    rng = np.random.default_rng(2024)
    gen = torch.Generator(device=dev).manual_seed(2024)
    clips = [ag.make_clip(rng, gen, lengths[i], dev) for i in picked]
    assert sum(c[0]["num_frames"] for c in clips) > 1500
    for _, gt in clips:
        gt.on(dev)

    kept = {}                                                    # clip position -> packed-run outputs (numpy)
    sample = {0, len(clips) - 1, 5, 23}                          # longest (121 frames), shortest (3 frames), two others
    assert clips[0][0]["num_frames"] == 121 and clips[-1][0]["num_frames"] == 3
    ev_pack = SceneGraphEvaluator_HIP(**kw)                      # the same loop with ONE evaluator call per pack
    ev_pack.register_container()
    for i in range(0, len(clips), PACK):
        group = clips[i:i + PACK]
        packed_pred = model(pack_clips([dict(c[0]) for c in group]))
        ev_pack.evaluate_packed([gt for _, gt in group], packed_pred)
        preds = unpack_predictions(packed_pred)
        for j, ((e, gt), p) in enumerate(zip(group, preds)):
            p.update(pair_idx=e["pair_idx"], im_idx=e["im_idx"], boxes=e["boxes"], labels=e["labels"], scores=e["scores"])
            ev_dev.evaluate_scene_graph(gt, p)
            ev_host.evaluate_scene_graph(gt.to_annotation(ev_host), p)
            if i + j in sample:
                kept[i + j] = {k: p[k].cpu().numpy() for k in OUT_KEYS}
    ev_dev.calculate_mean_recall(); ev_host.calculate_mean_recall(); ev_pack.calculate_mean_recall()
    torch.cuda.synchronize()

    # (1) identical evaluation: every recall list of every metric, the per-predicate collections and the mean-recall tables
    def same(x, y, path):
        if isinstance(x, dict):
            assert isinstance(y, dict) and set(x) == set(y), path
            for k in x:
                same(x[k], y[k], path + (k,))
        elif isinstance(x, (list, tuple)):
            assert len(x) == len(y), path
            for i, (u, v) in enumerate(zip(x, y)):
                same(u, v, path + (i,))
        else:
            assert float(x) == float(y), path
    assert set(ev_host.result_dict) == set(ev_dev.result_dict)
    same(ev_host.result_dict, ev_dev.result_dict, ())
    same(ev_host.result_dict, ev_pack.result_dict, ())           # one call per pack == one call per clip
    with pytest.raises(ValueError):                              # a pack whose ground truth misses a clip is refused
        ev_pack.evaluate_packed([gt for _, gt in clips[:PACK - 1]], model(pack_clips([dict(c[0]) for c in clips[:PACK]])))
    assert len(ev_host.result_dict["predcls_recall"][20]) == sum(c[0]["num_frames"] for c in clips)

    # (2) sampled clips vs the fp64 oracle, (3) packed == single
    for pos in sorted(sample):
        e = clips[pos][0]
        ref = orc.sttran_forward(_entry_to_numpy(e), sd, dtype=np.float64)
        single = model(dict(e))
        for k in OUT_KEYS:
            np.testing.assert_allclose(kept[pos][k], ref[k], atol=1e-3, rtol=0, err_msg=f"clip {pos} {k}")
            np.testing.assert_allclose(kept[pos][k], single[k].cpu().numpy(), atol=2e-5, rtol=0, err_msg=f"clip {pos} {k}")
