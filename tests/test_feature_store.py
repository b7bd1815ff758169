"""dets.npy / feat.npy reader: a committed fixture laid out the way the reference's writer lays it out
(tests/golden/gen_feature_store_fixture.py restates extract_bbox_features_ag.py:113-120 type for type), plus round
trips through this package's own writer."""
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


@pytest.mark.gpu
def test_loader_async_upload_of_the_reference_layout_fixture(golden_dir):
    """f-4's GPU leg on the COMMITTED fixture (files in the reference writer's layout, incl. the frame without
    detections): pinned staging + one asynchronous upload per clip on the side stream, slots reused, bytes on the device
    equal the stored ones"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root, exp, frames = _fixture(golden_dir)
    dirs = [os.path.join(root, f) for f in frames]
    n = [len(exp[f"class_{t}"]) for t in range(len(frames))]
    cat = lambda k: np.concatenate([exp[f"{k}_{t}"] for t in range(len(frames))])
    ld = ClipFeatureLoader(device="cuda:0")
    for rep in range(4):                                 # both slots, reused
        rec = ld.load(dirs)
        assert rec["features"].is_cuda and rec["boxes"].is_cuda and rec["boxes_per_frame"] == n
        torch.cuda.current_stream().wait_event(rec["ready"])
        assert rec["features"].cpu().numpy().tobytes() == cat("feat").tobytes()
        assert rec["boxes"][:, 1:].cpu().numpy().tobytes() == cat("rect").tobytes()
        np.testing.assert_array_equal(rec["boxes"][:, 0].cpu().numpy(), np.repeat(np.arange(len(frames)), n).astype(np.float32))
        np.testing.assert_array_equal(rec["classes"].numpy(), cat("class"))


def _fixture(golden_dir):
    root = os.path.join(golden_dir, "frame_features")
    exp = np.load(os.path.join(root, "expected.npz"))
    return root, exp, [str(f) for f in exp["frames"]]


def test_reference_layout_fixture_read_like_the_reference_reads_it(golden_dir):
    """lib/assign_pseudo_label.py:40-44: `np.load(dets_path, allow_pickle=True).tolist()` / `np.load(feat_path)`"""
    root, exp, frames = _fixture(golden_dir)
    for t, f in enumerate(frames):
        dets = np.load(os.path.join(root, f, "dets.npy"), allow_pickle=True).tolist()
        feat = np.load(os.path.join(root, f, "feat.npy"))
        assert isinstance(dets, list) and len(dets) == len(exp[f"class_{t}"])
        np.testing.assert_array_equal(feat, exp[f"feat_{t}"])
        for i, d in enumerate(dets):
            assert set(d) == {"class", "conf", "rect"}
            assert isinstance(d["class"], np.int64) and isinstance(d["conf"], np.float32)      # numpy scalars, not python
            assert d["rect"].dtype == np.float32 and d["rect"].shape == (4,)
            assert d["class"] == exp[f"class_{t}"][i] and d["conf"] == exp[f"conf_{t}"][i]
            np.testing.assert_array_equal(d["rect"], exp[f"rect_{t}"][i])


def test_loader_returns_the_fixture_bytes(golden_dir):
    """the loader on the reference-layout files (incl. the frame without detections, whose dets.npy is a float64
    array of shape (0,) because numpy saw an empty list) returns exactly the stored values"""
    root, exp, frames = _fixture(golden_dir)
    rec = ClipFeatureLoader(device=None).load([os.path.join(root, f) for f in frames])
    n = [len(exp[f"class_{t}"]) for t in range(len(frames))]
    assert rec["boxes_per_frame"] == n and rec["num_frames"] == len(frames)
    cat = lambda k: np.concatenate([exp[f"{k}_{t}"] for t in range(len(frames))])
    assert rec["features"].numpy().tobytes() == cat("feat").tobytes()
    assert rec["boxes"][:, 1:].numpy().tobytes() == cat("rect").tobytes()
    np.testing.assert_array_equal(rec["boxes"][:, 0].numpy(), np.repeat(np.arange(len(frames)), n).astype(np.float32))
    np.testing.assert_array_equal(rec["classes"].numpy(), cat("class"))
    assert rec["scores"].numpy().tobytes() == cat("conf").tobytes()


def test_own_writer_produces_the_reference_layout(tmp_path, golden_dir):
    """save_frame_features is byte-compatible with the fixture for the same values"""
    root, exp, frames = _fixture(golden_dir)
    for t, f in enumerate(frames):
        d = os.path.join(str(tmp_path), f)
        save_frame_features(d, exp[f"class_{t}"], exp[f"conf_{t}"], exp[f"rect_{t}"], exp[f"feat_{t}"])
        for name in ("dets.npy", "feat.npy"):
            a = np.load(os.path.join(d, name), allow_pickle=True)
            b = np.load(os.path.join(root, f, name), allow_pickle=True)
            assert a.dtype == b.dtype and a.shape == b.shape, (f, name)
        assert open(os.path.join(d, "feat.npy"), "rb").read() == open(os.path.join(root, f, "feat.npy"), "rb").read()
