"""State that survives between calls on one handle must never change a later call's result:
stale workspace contents (a NaN clip followed by a smaller clean clip), the cached index maps
(workspace growth, a failed call in between) and per-device launch state (two handles on two GPUs)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402

OUT_KEYS = ("attention_distribution", "spatial_distribution", "contacting_distribution")
CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]


def _model(sd, device="cuda:0"):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd.lib.sttran import STTran
    m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES,
               enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(device)
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    return m


def _cuda_entry(e, device="cuda:0"):
    return {k: (torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) and k != "frame_counts" else v)
            for k, v in e.items()}


def _run(m, e, device="cuda:0"):
    p = m(_cuda_entry(e, device))
    torch.cuda.synchronize()
    return {k: p[k].cpu().numpy() for k in OUT_KEYS}


@pytest.fixture(scope="module")
def weights():
    return syn.make_sttran_state_dict(7)


@pytest.mark.parametrize("poison", ["features", "union_feat", "spatial_masks"])
def test_nan_clip_does_not_contaminate_later_calls(poison, weights):
    """A clip with non-finite inputs (a corrupt feature file) gives non-finite outputs -- and nothing else.
    The next, smaller clip on the same handle must equal, bit for bit, its result on a fresh handle: the
    GEMM A loader reads 16 floats past column 1936 of every activation row (the K tail, multiplied by
    zero-padded weights), which must never be another call's leftovers."""
    big = syn.make_entry(301, [6, 5, 7, 6, 4])
    small = syn.make_entry(302, [2, 3, 1])
    clean = _run(_model(weights), small)
    m = _model(weights)
    bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in big.items()}
    bad[poison] = bad[poison].copy()
    bad[poison][...] = np.nan
    out = _run(m, bad)
    assert not np.isfinite(out["attention_distribution"]).any()
    again = _run(m, small)
    for k in OUT_KEYS:
        assert np.isfinite(again[k]).all(), k
        np.testing.assert_array_equal(again[k], clean[k], err_msg=k)
    # and with Inf left behind by a second poisoned call of yet another size
    bad2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in syn.make_entry(303, [3, 3, 3, 3]).items()}
    bad2[poison] = np.full_like(bad2[poison], np.inf)
    _run(m, bad2)
    again = _run(m, small)
    for k in OUT_KEYS:
        np.testing.assert_array_equal(again[k], clean[k], err_msg=k)


def test_reserve_between_forwards_keeps_results(weights):
    """forward -> reserve(larger) -> forward of the same clip: growing the workspace re-allocates (and zeroes) the
    device index buffer, so the cached layout must be rebuilt, not trusted."""
    e = syn.make_entry(311, [3, 1, 4, 2, 2])
    m = _model(weights)
    a = _run(m, e)
    m.reserve(5000, 6000)
    b = _run(m, e)
    m.reserve(20000, 24000)
    c = _run(m, e)
    for k in OUT_KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        np.testing.assert_array_equal(a[k], c[k], err_msg=k)


def test_failed_forward_does_not_poison_layout_cache(weights):
    """forward(A) ok, forward(B) fails after its counts were read (clip_num_frames that do not sum to the frames),
    forward(A) again: must not run A's device index maps with B's host-side offsets."""
    from nl_vsgg_amd._native import SttranError
    A = syn.make_entry(321, [4, 2, 5, 3])
    m = _model(weights)
    a = _run(m, A)
    B = dict(_cuda_entry(syn.make_entry(322, [3, 3, 2])))
    B["clip_num_frames"] = np.array([2, 2], dtype=np.int32)            # 4 frames claimed, 3 present
    with pytest.raises(SttranError) as ei:
        m(B)
    assert ei.value.code == 1
    b = _run(m, A)
    bad = dict(_cuda_entry(A)); bad["frame_counts"] = np.array([4, 2, 5, 4], dtype=np.int32)   # sums to 15 != 14
    with pytest.raises(SttranError):
        m(bad)
    c = _run(m, A)
    for k in OUT_KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        np.testing.assert_array_equal(a[k], c[k], err_msg=k)


def test_two_handles_on_two_devices(weights):
    """sttran_create accepts any device ordinal: the > 64 KB dynamic-LDS limits and the CU count the tile planner
    uses are per-device state (a process-wide 'already configured' flag would skip device 1)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    e = syn.make_entry(331, [5, 3, 6, 2])
    a = _run(_model(weights, "cuda:0"), e, "cuda:0")
    with torch.cuda.device(1):
        m1 = _model(weights, "cuda:1")
        p = m1(_cuda_entry(e, "cuda:1"))
        torch.cuda.synchronize(1)
        b = {k: p[k].cpu().numpy() for k in OUT_KEYS}
    for k in OUT_KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_strict_inputs_refuses_hidden_copy(weights):
    """a non-contiguous / wrong-dtype / host tensor is converted (with a warning for big ones) by default and
    refused under strict_inputs"""
    e = _cuda_entry(syn.make_entry(341, [3, 2, 4]))
    m = _model(weights)
    ref = {k: m(dict(e))[k].cpu().numpy() for k in OUT_KEYS}
    nc = dict(e)
    nc["union_feat"] = e["union_feat"].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)     # same values, NHWC strides
    assert not nc["union_feat"].is_contiguous()
    got = m(dict(nc))
    for k in OUT_KEYS:
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    m.strict_inputs = True
    with pytest.raises(ValueError, match="union_feat"):
        m(dict(nc))
    host = dict(e); host["features"] = e["features"].cpu()
    with pytest.raises(ValueError, match="features"):
        m(host)
    dbl = dict(e); dbl["spatial_masks"] = e["spatial_masks"].double()
    with pytest.raises(ValueError, match="spatial_masks"):
        m(dbl)
    got = m(dict(e))                                        # the well-formed entry still runs
    for k in OUT_KEYS:
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])


def test_shape_errors_like_the_reference(weights):
    from nl_vsgg_amd.lib.sttran import STTran
    e = _cuda_entry(syn.make_entry(342, [2, 2]))
    m = _model(weights)
    bad = dict(e); bad["spatial_masks"] = e["spatial_masks"][:, :, :26, :26].contiguous()
    with pytest.raises(ValueError):
        m(bad)
    bad = dict(e); bad["union_feat"] = e["union_feat"][:, :1024].contiguous()
    with pytest.raises(ValueError):
        m(bad)
    # sgdet: an entry that already went through forward carries the [B, 37] logits in `distribution`
    sg = STTran(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES,
                enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    sg.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()}, strict=False)
    es = _cuda_entry(syn.make_entry(343, [2, 3], mode="sgdet", im_idx_dtype=np.int64))
    out = sg(es)
    assert out["distribution"].shape[1] == 37
    with pytest.raises(ValueError, match="already forwarded"):
        sg(out)


def test_out_of_range_index_is_an_index_error(weights):
    """check_indices defaults to True: the torch indexing of lib/sttran.py:381-393 raises IndexError"""
    e = _cuda_entry(syn.make_entry(344, [2, 2]))
    m = _model(weights)
    assert m.check_indices is True
    bad = dict(e); bad["pair_idx"] = e["pair_idx"].clone(); bad["pair_idx"][1, 1] = 10_000
    with pytest.raises(IndexError):
        m(bad)
    m.check_indices = False
    m(dict(bad))                                            # clamped, flagged on the device ...
    with pytest.raises(IndexError):
        m.sync_check()                                      # ... and reported when asked
    m.sync_check()                                          # the flag is cleared by the report


def test_dsg_detr_is_enqueue_only_and_capturable():
    """DSG-DETR builds its class sequences on the device (clips of <= 480 pairs): with check_indices off the forward only
    enqueues -- it can be captured into a HIP graph and replayed on new labels --, and an out-of-range label is reported
    the way STTran reports it (clamped, flagged, IndexError at the next check)"""
    from nl_vsgg_amd.lib.dsg_detr import STTran as DSG
    sd = syn.make_dsg_detr_state_dict(7)
    m = DSG(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=CLASSES).to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    e = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v)
         for k, v in syn.make_entry(77, [3, 2, 4, 1], mode="sgdet", im_idx_dtype=np.int64).items()}
    keep = {k: e[k].clone() for k in ("labels", "distribution")}
    want = {k: v.clone() for k, v in m(dict(e)).items() if k.endswith("_distribution")}
    m.check_indices = False
    static = dict(e)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(dict(static)); m(dict(static))
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(dict(static))
    g.replay()
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(out[k], want[k]), k
    # new labels in the captured input tensor: the replay regroups the pairs on the device
    perm = torch.randperm(36, device="cuda") + 1
    static["labels"].copy_(torch.where(keep["labels"] == 1, keep["labels"], perm[(keep["labels"] - 1).clamp(min=0)]))
    g.replay()
    torch.cuda.synchronize()
    fresh = dict(e); fresh["labels"] = static["labels"].clone(); fresh["distribution"] = keep["distribution"].clone()
    m2 = m(dict(fresh))
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(out[k], m2[k]), k
    assert not torch.equal(out["attention_distribution"], want["attention_distribution"])
    # out-of-range label: clamped + flagged, raised by the check
    m.check_indices = True
    bad = dict(e); bad["labels"] = keep["labels"].clone(); bad["labels"][2] = 99; bad["distribution"] = keep["distribution"].clone()
    with pytest.raises(IndexError):
        m(bad)


def test_by_pointer_batch_is_enqueue_only_and_capturable(weights):
    """a batch handed over as per-clip pointer tables: with the frame counts given and check_indices off the forward only
    enqueues (the chunk table is re-uploaded only when a call's pointers change), so a step that packs by pointer can be
    captured into a HIP graph and replayed on new feature VALUES in the same tensors"""
    from nl_vsgg_amd.lib.sttran import pack_clips
    m = _model(weights)
    m.check_indices = False
    clips = [_cuda_entry(syn.make_entry(700 + i, c)) for i, c in enumerate([[3, 2, 4], [5, 1], [2, 2, 2, 2]])]
    want = {k: v.clone() for k, v in m(pack_clips(clips, copy=False)).items() if k in OUT_KEYS}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(pack_clips(clips, copy=False)); m(pack_clips(clips, copy=False))
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(pack_clips(clips, copy=False))
    g.replay()
    torch.cuda.synchronize()
    for k in OUT_KEYS:
        assert torch.equal(out[k], want[k]), k
    # new values in the SAME tensors of one clip: the replay reads them where they are
    clips[1]["features"].mul_(0.5)
    g.replay()
    torch.cuda.synchronize()
    fresh = m(pack_clips(clips, copy=False))
    torch.cuda.synchronize()
    for k in OUT_KEYS:
        assert torch.equal(out[k], fresh[k]), k
    assert not torch.equal(out["attention_distribution"], want["attention_distribution"])
    m.sync_check()


def test_interpreter_exit_with_live_models_is_clean():
    """models (and a captured graph) still alive at interpreter exit: the atexit hook destroys the native handles while the
    HIP runtime is intact, __del__ stays out of module teardown -- the process exits 0"""
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, sys\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from nl_vsgg_amd.lib import synthetic as syn\n"
        "from nl_vsgg_amd.lib.sttran import STTran, _LIVE\n"
        "C = ['__background__'] + [f'c{i}' for i in range(36)]\n"
        "sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}\n"
        "keep = []\n"
        "for mode in ('predcls', 'sgdet'):\n"
        "    m = STTran(mode=mode, attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=C,\n"
        "               enc_layer_num=1, dec_layer_num=3, transformer_mode='wk', is_wks=True, feat_dim=2048).to('cuda:0')\n"
        "    m.load_state_dict(sd, strict=False)\n"
        "    e = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != 'frame_counts' else v)\n"
        "         for k, v in syn.make_entry(5, [3, 2, 4], mode=mode).items()}\n"
        "    keep.append((m, m(e)))\n"
        "assert len(_LIVE) == 2\n"
        "print('LIVE', len(_LIVE))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LIVE 2" in r.stdout, f"rc={r.returncode}\n{r.stdout[-1000:]}\n{r.stderr[-3000:]}"
