"""dets.npy / feat.npy reader: round trip through the reference's on-disk format."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from nl_vsgg_amd.lib.feature_store import ClipFeatureLoader, read_frame, save_frame_features


def _write_clip(tmp, counts, seed=0):
    rng = np.random.default_rng(seed)
    dirs, ref = [], []
    for t, n in enumerate(counts):
        d = os.path.join(tmp, f"vid.mp4/{t:06d}.png")
        cls = rng.integers(1, 37, n); conf = rng.random(n).astype(np.float32)
        xy = rng.random((n, 2)).astype(np.float32) * 300; wh = rng.random((n, 2)).astype(np.float32) * 100 + 5
        rect = np.concatenate([xy, xy + wh], 1); feat = rng.standard_normal((n, 2048)).astype(np.float32)
        save_frame_features(d, cls.tolist(), conf.tolist(), rect, feat)
        dirs.append(d); ref.append((cls, conf, rect, feat))
    return dirs, ref


def test_format_matches_reference_reader(tmp_path):
    """the files are readable exactly the way lib/assign_pseudo_label.py:40-44 reads them"""
    dirs, ref = _write_clip(str(tmp_path), [3])
    dets = np.load(os.path.join(dirs[0], "dets.npy"), allow_pickle=True).tolist()
    feat = np.load(os.path.join(dirs[0], "feat.npy"))
    assert isinstance(dets, list) and set(dets[0]) == {"class", "conf", "rect"}
    np.testing.assert_array_equal(feat, ref[0][3])


@pytest.mark.parametrize("counts", [[3, 0, 5, 1], [1]])
def test_loader_round_trip_cpu(tmp_path, counts):
    dirs, ref = _write_clip(str(tmp_path), counts, seed=len(counts))
    rec = ClipFeatureLoader(device=None).load(dirs)
    assert rec["boxes_per_frame"] == counts and rec["num_frames"] == len(counts)
    o = 0
    for t, (cls, conf, rect, feat) in enumerate(ref):
        n = len(cls)
        np.testing.assert_array_equal(rec["features"][o:o + n].numpy(), feat)
        np.testing.assert_array_equal(rec["boxes"][o:o + n, 1:].numpy(), rect)
        assert (rec["boxes"][o:o + n, 0] == t).all()
        np.testing.assert_array_equal(rec["classes"][o:o + n].numpy(), cls)
        o += n


@pytest.mark.gpu
def test_loader_async_upload(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dirs, ref = _write_clip(str(tmp_path), [4, 2, 6])
    ld = ClipFeatureLoader(device="cuda:0")
    for _ in range(3):                                   # slot reuse
        rec = ld.load(dirs)
        torch.cuda.current_stream().wait_event(rec["ready"])
        got = rec["features"].cpu().numpy()
        np.testing.assert_array_equal(got, np.concatenate([r[3] for r in ref]))
