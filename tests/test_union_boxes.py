"""draw_union_boxes: the numpy restatement (used to build synthetic entries) and the HIP kernel against
the reference's Cython output (tests/golden/draw_union_boxes.npz, made by gen_golden_eval.py)."""
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn


def test_numpy_restatement_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "draw_union_boxes.npz"))
    out = syn.union_box_masks(g["pair_rois"], 27)
    np.testing.assert_array_equal(out, g["masks"])          # float32, same operation order: bit-exact


@pytest.mark.gpu
def test_hip_kernel_matches_reference(golden_dir):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.union_boxes import union_boxes_and_masks
    g = np.load(os.path.join(golden_dir, "draw_union_boxes.npz"))
    rois = g["pair_rois"]
    P = rois.shape[0]
    boxes = np.zeros((2 * P, 5), dtype=np.float32)
    boxes[0::2, 1:] = rois[:, :4]
    boxes[1::2, 1:] = rois[:, 4:]
    boxes[:, 0] = np.repeat(np.arange(P), 2)
    pair = np.stack([np.arange(P) * 2, np.arange(P) * 2 + 1], axis=1).astype(np.int64)
    ub, masks = union_boxes_and_masks(torch.from_numpy(boxes).cuda(), torch.from_numpy(pair).cuda(),
                                      torch.arange(P, dtype=torch.float32).cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(masks.cpu().numpy(), g["masks"] - np.float32(0.5), atol=1e-6, rtol=0)
    exp = np.concatenate([np.arange(P, dtype=np.float32)[:, None], np.minimum(rois[:, :2], rois[:, 4:6]),
                          np.maximum(rois[:, 2:4], rois[:, 6:8])], axis=1)
    np.testing.assert_array_equal(ub.cpu().numpy(), exp)
