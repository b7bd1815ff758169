"""Writes tests/golden/frame_features/: a three-frame clip in the reference's on-disk VinVL feature format, laid out
exactly as the reference writer lays it out, plus the arrays a reader must return (expected.npz).

The writer (`NL-VSGG/data_preprocess/extract_bbox_features_ag.py:108-120`) cannot be imported here (it needs
maskrcnn_benchmark and a detector checkpoint), so its save statements are restated with the same Python / numpy types:

    cls_info  = BoxList.extra_fields['labels'].numpy()        -> int64   [n]
    conf_info = BoxList.extra_fields['scores'].numpy()        -> float32 [n]
    bbox_info = BoxList.bbox.numpy()                          -> float32 [n, 4]  (x1, y1, x2, y2)
    feat_info = BoxList.extra_fields['box_features'].numpy()  -> float32 [n, 2048]
    per_img_info = [{'class': cls_info[i], 'conf': conf_info[i], 'rect': bbox_info[i]} for i in range(n)]
    np.save(f"{dir_name}/dets.npy", per_img_info, allow_pickle=True)     # a LIST of dicts with numpy-scalar fields
    np.save(f"{dir_name}/feat.npy", feat_info)

i.e. dets.npy is a pickled object array built by numpy from a Python list (an EMPTY list becomes a float64 array of
shape (0,)), `class` is a numpy.int64 scalar, `conf` a numpy.float32 scalar and `rect` a float32[4] row view.
The reader this stands in for is `lib/assign_pseudo_label.py:27-45::load_feature`
(`np.load(dets_path, allow_pickle=True).tolist()`, `np.load(feat_path)`).

Run from the repo root:  python tests/golden/gen_feature_store_fixture.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "frame_features")
FRAMES = [("001YG.mp4/000089.png", 3), ("001YG.mp4/000093.png", 0), ("001YG.mp4/000264.png", 2)]


def main():
    rng = np.random.default_rng(20261003)
    expected = {}
    for t, (frame_name, n) in enumerate(FRAMES):
        cls_info = rng.integers(1, 1595, n).astype(np.int64)            # VinVL / OpenImages-style class ids
        conf_info = rng.random(n).astype(np.float32)
        xy = (rng.random((n, 2)) * 300).astype(np.float32)
        bbox_info = np.concatenate([xy, xy + (rng.random((n, 2)) * 150 + 8).astype(np.float32)], axis=1).astype(np.float32)
        feat_info = rng.standard_normal((n, 2048)).astype(np.float32)
        # ---- the reference's statements (extract_bbox_features_ag.py:113-120), same types ----
        per_img_info = []
        for idx_per_box in range(n):
            per_img_info.append({'class': cls_info[idx_per_box], 'conf': conf_info[idx_per_box], 'rect': bbox_info[idx_per_box]})
        dir_name = os.path.join(ROOT, frame_name)
        os.makedirs(dir_name, exist_ok=True)
        np.save(f"{dir_name}/dets.npy", per_img_info, allow_pickle=True)
        np.save(f"{dir_name}/feat.npy", feat_info)
        # ---------------------------------------------------------------------------------------
        expected[f"class_{t}"], expected[f"conf_{t}"] = cls_info, conf_info
        expected[f"rect_{t}"], expected[f"feat_{t}"] = bbox_info, feat_info
    expected["frames"] = np.array([f for f, _ in FRAMES])
    np.savez(os.path.join(ROOT, "expected.npz"), **expected)
    print("wrote", ROOT)


if __name__ == "__main__":
    main()
