#!/usr/bin/env python3
"""Known-answer vectors for the evaluator: runs the REFERENCE `lib/evaluation_recall.py`
(build container only) on the golden model outputs + seeded synthetic ground truth, and stores the
recall lists it produces as tests/golden/eval_<case>.json (data only).

The reference's Cython IoU (`lib/fpn/box_intersections_cpu/bbox.pyx`) is compiled out-of-tree into a
temp dir with `numpy.float = float` (it predates numpy 1.24)."""
import importlib
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

CASES = {"uniform_16x12": "predcls", "ragged_5": "predcls", "sgdet_ragged": "sgdet"}
# round 2: ground truth the matcher has to work for (synthetic.make_gt_annotation_hard: every frame present -- also
# those without a predicted pair --, boxes jittered around the 0.5 IoU line, some classes flipped).
# name -> (model fixture, mode, jitter)
HARD_CASES = {"hard_empty_frames": ("empty_frames", "predcls", 6.0), "hard_sgdet_empty_frames": ("sgdet_empty_frames", "sgdet", 6.0),
              "hard_uniform_16x12": ("uniform_16x12", "predcls", 8.0), "hard_sgdet_16x12": ("sgdet_16x12", "sgdet", 8.0)}
OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]
SPA = [f"spa{i}" for i in range(6)]
CON = [f"con{i}" for i in range(17)]


def build_bbox():
    np.float = float  # noqa
    tmp = tempfile.mkdtemp(prefix="refbbox_")
    src = os.path.join(REF, "lib/fpn/box_intersections_cpu/bbox.pyx")
    setup = os.path.join(tmp, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                f"setup(ext_modules=cythonize(Extension('bbox', [r'{src}'], include_dirs=[numpy.get_include()]),"
                f" language_level=2, build_dir=r'{tmp}/b'))\n")
    subprocess.run([sys.executable, setup, "build_ext", "--build-lib", tmp, "--build-temp", tmp + "/t"],
                   check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, tmp)
    bbox = importlib.import_module("bbox")
    import lib  # noqa
    pk = types.ModuleType("lib.fpn.box_intersections_cpu"); pk.__path__ = []
    sys.modules[pk.__name__] = pk
    sys.modules["lib.fpn.box_intersections_cpu.bbox"] = bbox
    sys.modules["h5py"] = types.ModuleType("h5py")


def build_draw():
    """compile the reference's draw_rectangles.pyx out-of-tree and return draw_union_boxes"""
    tmp = tempfile.mkdtemp(prefix="refdraw_")
    src = os.path.join(REF, "lib/draw_rectangles/draw_rectangles.pyx")
    setup = os.path.join(tmp, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                f"setup(ext_modules=cythonize(Extension('draw_rectangles', [r'{src}'], include_dirs=[numpy.get_include()]),"
                f" language_level=2, build_dir=r'{tmp}/b'))\n")
    subprocess.run([sys.executable, setup, "build_ext", "--build-lib", tmp, "--build-temp", tmp + "/t"],
                   check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, tmp)
    return importlib.import_module("draw_rectangles").draw_union_boxes


def golden_draw():
    draw = build_draw()
    e = syn.make_entry(301, [3, 2, 4])
    pi = e["pair_idx"]
    rois = np.concatenate([e["boxes"][pi[:, 0], 1:], e["boxes"][pi[:, 1], 1:]], axis=1).astype(np.float32)
    # a few hand-made edge cases: identical boxes, nested box, touching boxes
    extra = np.array([[10, 10, 50, 60, 10, 10, 50, 60], [0, 0, 100, 100, 25, 30, 40, 45],
                      [0, 0, 10, 10, 10, 0, 20, 10]], dtype=np.float32)
    rois = np.concatenate([rois, extra], axis=0)
    out = draw(rois, 27)
    np.savez_compressed(os.path.join(HERE, "draw_union_boxes.npz"), pair_rois=rois, masks=out.astype(np.float32))
    print("draw_union_boxes", out.shape, float(out.min()), float(out.max()))


def run_reference_evaluator(RefEval, mode, e, gt, g):
    for fr in gt:
        for o in fr[1:]:
            for k in ("attention_relationship", "spatial_relationship", "contacting_relationship"):
                o[k] = torch.from_numpy(np.asarray(o[k]))
    pred = {k: torch.from_numpy(e[k]) for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        pred[k] = torch.from_numpy(g[k])
    pred["pred_labels"], pred["pred_scores"] = pred["labels"], pred["scores"]
    ev = RefEval(mode=mode, AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON,
                 AG_attention_predicates=ATT, AG_spatial_predicates=SPA, AG_contacting_predicates=CON,
                 iou_threshold=0.5, constraint="with")
    ev.register_container()
    ev.evaluate_scene_graph(gt, pred)
    ev.calculate_mean_recall()
    out = {}
    for key, val in ev.result_dict.items():
        if key.endswith("_collect"):
            continue
        out[key] = {str(k): (v if isinstance(v, (int, float)) else [float(x) for x in v]) for k, v in val.items()}
    return ev, out


def main():
    golden_draw()
    build_bbox()
    from lib.evaluation_recall import SceneGraphEvaluator as RefEval
    for case, mode in CASES.items():
        g = np.load(os.path.join(HERE, f"sttran_{case}.npz"))
        e = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                           im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
        gt = syn.make_gt_annotation(1000 + int(g["entry_seed"]), e)
        ev, out = run_reference_evaluator(RefEval, mode, e, gt, g)
        with open(os.path.join(HERE, f"eval_{case}.json"), "w") as f:
            json.dump({"mode": mode, "gt_seed": 1000 + int(g["entry_seed"]), "result_dict": out}, f, indent=0)
        print(case, {k: round(float(np.mean(v)), 4) for k, v in ev.result_dict[mode + "_recall"].items()})
    for case, (fixture, mode, jitter) in HARD_CASES.items():
        g = np.load(os.path.join(HERE, f"sttran_{fixture}.npz"))
        e = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                           im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
        gt = syn.make_gt_annotation_hard(2000 + int(g["entry_seed"]), e, jitter=jitter)
        assert len(gt) == int(e["num_frames"])
        ev, out = run_reference_evaluator(RefEval, mode, e, gt, g)
        with open(os.path.join(HERE, f"eval_{case}.json"), "w") as f:
            json.dump({"mode": mode, "fixture": fixture, "gt": "hard", "jitter": jitter, "gt_seed": 2000 + int(g["entry_seed"]),
                       "result_dict": out}, f, indent=0)
        print(case, {k: round(float(np.mean(v)), 4) for k, v in ev.result_dict[mode + "_recall"].items()},
              {k: round(float(np.mean(v)), 4) for k, v in ev.result_dict[mode + "_recall_nogc"].items()})


if __name__ == "__main__":
    main()
