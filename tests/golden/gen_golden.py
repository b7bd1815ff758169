#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
It imports the reference's `lib/sttran.py`, `lib/transformer.py` (SURVEY Appendix B recipe),
feeds them the seeded inputs/weights of `nl-vsgg_amd/lib/synthetic.py`, and stores the outputs
(data only) as small .npz files.  Nothing of the reference's source is copied.

    python tests/golden/gen_golden.py            # rewrites tests/golden/sttran_*.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

WEIGHT_SEED = 7

# name -> (entry seed, pairs per frame, mode, dump stage tensors?)
CASES = {
    "uniform_3x2":   (101, [1, 1, 1], "predcls", True),
    "ragged_5":      (102, [3, 1, 4, 2, 2], "predcls", True),
    "empty_frames":  (103, [2, 0, 3, 0, 0, 2], "predcls", True),
    "two_frames":    (104, [2, 3], "predcls", True),
    "uniform_16x12": (105, [11] * 16, "predcls", False),
    "sgdet_ragged":  (106, [2, 4, 1, 3], "sgdet", True),
    "uniform_64x36": (107, [35] * 64, "predcls", False),      # BASELINE.json configs[3] at full size
    # the longest clip of the AG test split: 121 frames, 0..6 pairs per frame (empty frames and windows inside)
    "ragged_121": (108, [int(c) for c in np.random.default_rng(121).integers(0, 7, 121)], "predcls", False),
    "sgdet_16x12": (109, [11] * 16, "sgdet", False),
    # round 2: the clip STARTS with frames that hold no pair (the j == 0 scatter of lib/transformer.py:180 is empty)
    "leading_empty": (110, [0, 0, 3, 0, 2, 1], "predcls", True),
    "sgdet_empty_frames": (111, [0, 2, 0, 3, 1], "sgdet", True),
}


def _stub_modules():
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ("fasterRCNN", "fasterRCNN.lib", "fasterRCNN.lib.model"):
        pkg(n)
    rl = types.ModuleType("fasterRCNN.lib.model.roi_layers")

    class ROIAlign(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    rl.ROIAlign = ROIAlign
    rl.nms = lambda *a, **k: None
    sys.modules[rl.__name__] = rl
    dr = types.ModuleType("lib.draw_rectangles.draw_rectangles")
    dr.draw_union_boxes = lambda *a, **k: None
    import lib  # the reference's package
    p = types.ModuleType("lib.draw_rectangles"); p.__path__ = []
    sys.modules["lib.draw_rectangles"] = p
    sys.modules[dr.__name__] = dr
    eb = types.ModuleType("lib.extract_bbox_features")
    eb.extract_feature_given_bbox_base_feat_torch = lambda *a, **k: None
    sys.modules[eb.__name__] = eb
    bi = types.ModuleType("lib.fpn.box_intersections_cpu.bbox")
    bi.bbox_overlaps = bi.bbox_intersections = lambda *a, **k: None
    p2 = types.ModuleType("lib.fpn.box_intersections_cpu"); p2.__path__ = []
    sys.modules[p2.__name__] = p2
    sys.modules[bi.__name__] = bi


def build_reference_model(mode, sd_np):
    _stub_modules()
    import lib.word_vectors as wv
    wv.obj_edge_vectors = lambda names, **k: torch.zeros(len(names), 200)
    import lib.sttran as rs
    import lib.transformer as rt
    rs.obj_edge_vectors = wv.obj_edge_vectors
    classes = ["__background__"] + [f"c{i}" for i in range(36)]
    m = rs.STTran(mode=mode, attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                  obj_classes=classes, enc_layer_num=1, dec_layer_num=3, transformer_mode="wk",
                  is_wks=True, feat_dim=2048)
    m.eval()
    # torch>=2 rejects transformer_wk's int key-padding mask; the two classes
    # share attribute names, so swap in lib/transformer.py::transformer (bool / -inf semantics).
    m.glocal_transformer.__class__ = rt.transformer
    ref_sd = m.state_dict()
    assert set(ref_sd.keys()) == set(sd_np.keys()), (set(ref_sd) ^ set(sd_np))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd_np[k].shape), (k, v.shape, sd_np[k].shape)
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return m


def run_case(model, name, seed, counts, mode, dump):
    e_np = syn.make_entry(seed, counts, mode=mode,
                          im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    entry = {k: torch.from_numpy(v) for k, v in e_np.items() if isinstance(v, np.ndarray) and k != "frame_counts"}
    cnt = np.asarray(counts)
    T = len(counts)
    # the reference sees only frames up to the last non-empty one (b = im_idx[-1] + 1)
    stages = {}
    hooks = []
    tr = model.glocal_transformer

    def want_mask():
        l = int(cnt.max())
        b = int(e_np["im_idx"][-1]) + 1
        mk = np.ones((b - 1, 2 * l), dtype=bool)
        for j in range(b - 1):
            mk[j, : cnt[j] + cnt[j + 1]] = False
        return mk

    def dec_hook(mod, args):
        got = args[1].numpy()
        exp = want_mask()
        if not np.array_equal(got, exp):
            raise RuntimeError(f"{name}: reference row-sum mask != counts mask (SURVEY fact 6); fixture invalid")

    hooks.append(tr.global_attention.register_forward_pre_hook(dec_hook))

    def grab(key):
        def f(mod, args, out):
            stages[key] = (out[0] if isinstance(out, tuple) else out).detach().numpy().copy()
        return f

    def grab_in(key):
        def f(mod, args, kwargs):
            stages[key] = kwargs["features"].detach().numpy().copy()
        return f

    hooks.append(tr.register_forward_pre_hook(grab_in("rel_features"), with_kwargs=True))
    hooks.append(tr.register_forward_hook(grab("global_output")))
    hooks.append(tr.local_attention.register_forward_hook(grab("local_padded")))
    for i, layer in enumerate(tr.global_attention.layers):
        hooks.append(layer.register_forward_hook(grab(f"dec{i}_padded")))
    with torch.no_grad():
        pred = model(entry)
    for h in hooks:
        h.remove()
    out = {
        "attention_distribution": pred["attention_distribution"].numpy(),
        "spatial_distribution": pred["spatial_distribution"].numpy(),
        "contacting_distribution": pred["contacting_distribution"].numpy(),
        "pairs_per_frame": cnt.astype(np.int64),
        "entry_seed": np.int64(seed), "weight_seed": np.int64(WEIGHT_SEED),
    }
    if mode == "sgdet":
        out["distribution"] = pred["distribution"].numpy()
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        assert np.isfinite(out[k]).all(), (name, k)
    if dump:
        b = int(e_np["im_idx"][-1]) + 1
        l = int(cnt.max())
        out["rel_features"] = stages["rel_features"]
        out["global_output"] = stages["global_output"]
        # un-pad the encoder output (lib/transformer.py:145): [l,b,D] -> rows in pair order
        lp = stages["local_padded"]
        out["local_output"] = np.concatenate([lp[: cnt[t], t] for t in range(b)], axis=0)
        for i in range(3):
            dp = stages[f"dec{i}_padded"]                    # [2l, b-1, D]
            out[f"decoder_layer{i}"] = np.concatenate(
                [dp[: cnt[j] + cnt[j + 1], j] for j in range(b - 1)], axis=0)
    else:
        out["rel_features_head"] = stages["rel_features"][:4]
    np.savez_compressed(os.path.join(HERE, f"sttran_{name}.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print(f"{name}: P={len(e_np['im_idx'])} wrote sttran_{name}.npz")


DSG_CASES = {
    "dsgdetr_4x3": (201, [2, 2, 2, 2]),
    "dsgdetr_ragged": (202, [3, 1, 4, 2, 5]),
    "dsgdetr_16x12": (203, [11] * 16),
    # box rows stored in a random order: the subject numbers of a class sequence are no longer ascending, and
    # lib/dsg_detr.py:551-555 hands out position indices by POSITION in the sequence (sorted counts), not per subject
    "dsgdetr_shuffled_boxes": (204, [3, 1, 4, 2, 5, 3], 9),
    "dsgdetr_empty_frames": (205, [0, 2, 0, 3, 1, 0, 2]),
}


def build_reference_dsg(sd_np):
    _stub_modules()
    tv = types.ModuleType("torchvision"); tv.__path__ = []
    ops = types.ModuleType("torchvision.ops"); ops.__path__ = []
    bx = types.ModuleType("torchvision.ops.boxes"); bx.box_area = lambda b: None
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": bx})
    import lib.word_vectors as wv
    wv.obj_edge_vectors = lambda names, **k: torch.zeros(len(names), 200)
    import lib.dsg_detr as rd
    rd.obj_edge_vectors = wv.obj_edge_vectors
    torch.Tensor.cuda = lambda self, *a, **k: self          # hard-coded .cuda() calls (lib/dsg_detr.py:542,559)
    classes = ["__background__"] + [f"c{i}" for i in range(36)]
    m = rd.STTran(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=classes)
    m.eval()
    ref_sd = m.state_dict()
    for k, v in sd_np.items():
        assert k in ref_sd and tuple(ref_sd[k].shape) == tuple(np.asarray(v).shape), k
    unused = [k for k in ref_sd if k not in sd_np and "num_batches_tracked" not in k]
    assert all(k.startswith("object_classifier.encoder_tran.") or k == "object_classifier.positional_encoder.pe"
               for k in unused), unused
    # the reference's own sinusoid table must agree with the portable generator's
    assert np.abs(ref_sd["positional_encoder.pe"].numpy() - sd_np["positional_encoder.pe"]).max() < 1e-4  # float32 sin(399*x)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=False)
    return m


def run_dsg_case(model, name, seed, counts, shuffle_seed=None):
    e_np = syn.make_entry(seed, counts, mode="sgdet", im_idx_dtype=np.int64)
    if shuffle_seed is not None:
        e_np = syn.shuffle_boxes(e_np, shuffle_seed)
    entry = {k: torch.from_numpy(v) for k, v in e_np.items() if isinstance(v, np.ndarray) and k != "frame_counts"}
    grabbed = {}
    h = model.local_transformer.register_forward_hook(lambda m, a, o: grabbed.__setitem__("local_padded", o.detach().numpy().copy()))
    with torch.no_grad():
        pred = model(entry)
    h.remove()
    out = {k: pred[k].numpy() for k in ("attention_distribution", "spatial_distribution", "contacting_distribution",
                                        "distribution")}
    cnt = np.asarray(counts)
    # one padded row per NON-EMPTY frame (lib/dsg_detr.py:537-538 loops over im_indices.unique())
    lo = np.concatenate([grabbed["local_padded"][r, :n] for r, n in enumerate(cnt[cnt > 0])], axis=0)
    if lo.shape[0] <= 64:
        out["local_output"] = lo
    else:
        out["local_output_head"] = lo[:4]                      # keep the larger fixtures small
    out.update(pairs_per_frame=cnt.astype(np.int64), entry_seed=np.int64(seed), weight_seed=np.int64(WEIGHT_SEED))
    if shuffle_seed is not None:
        out["box_shuffle_seed"] = np.int64(shuffle_seed)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        assert np.isfinite(out[k]).all(), (name, k)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: P={len(e_np['im_idx'])} wrote {name}.npz")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = syn.make_sttran_state_dict(WEIGHT_SEED)
    models = {}
    for name, (seed, counts, mode, dump) in CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        if mode not in models:
            models[mode] = build_reference_model(mode, sd)
        run_case(models[mode], name, seed, counts, mode, dump)
    dsg = None
    for name, case in DSG_CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        if dsg is None:
            dsg = build_reference_dsg(syn.make_dsg_detr_state_dict(WEIGHT_SEED))
        run_dsg_case(dsg, name, *case)


if __name__ == "__main__":
    main()
