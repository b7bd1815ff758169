"""The numpy oracle (oracle/sttran_oracle.py) against golden vectors produced by the imported
reference (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn
from oracle import sttran_oracle as orc

CASES = ["uniform_3x2", "ragged_5", "empty_frames", "two_frames", "sgdet_ragged", "uniform_16x12", "ragged_121",
         "sgdet_16x12", "leading_empty", "sgdet_empty_frames"]
TOL = 2e-5          # fp32 oracle vs fp32 torch-CPU reference: different summation orders only


@pytest.fixture(scope="module")
def weights():
    return {}


def _weights(cache, seed):
    if seed not in cache:
        cache[seed] = syn.make_sttran_state_dict(seed)
    return cache[seed]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name, golden_dir, weights):
    g = np.load(os.path.join(golden_dir, f"sttran_{name}.npz"))
    mode = "sgdet" if "distribution" in g.files else "predcls"
    sd = _weights(weights, int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                           im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    stages = {}
    out = orc.sttran_forward(entry, sd, mode=mode, stages=stages)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        np.testing.assert_allclose(out[k], g[k], atol=TOL, rtol=0, err_msg=k)
    if mode == "sgdet":
        np.testing.assert_allclose(out["distribution"], g["distribution"], atol=TOL, rtol=0)
    if "rel_features" in g.files:
        for k in ("rel_features", "local_output", "decoder_layer0", "decoder_layer1",
                  "decoder_layer2", "global_output"):
            np.testing.assert_allclose(stages[k], g[k], atol=5e-5, rtol=0, err_msg=k)
    else:
        np.testing.assert_allclose(stages["rel_features"][:4], g["rel_features_head"], atol=5e-5, rtol=0)


@pytest.mark.skipif(os.environ.get("STTRAN_SLOW_TESTS") != "1", reason="~1 min and 2 GB: set STTRAN_SLOW_TESTS=1")
def test_oracle_full_size_64x36(weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "sttran_uniform_64x36.npz"))
    sd = _weights(weights, int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist())
    out = orc.sttran_forward(entry, sd)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        np.testing.assert_allclose(out[k], g[k], atol=5e-5, rtol=0, err_msg=k)


def test_oracle_fp64_agrees_with_fp32(weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "sttran_ragged_5.npz"))
    sd = _weights(weights, int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist())
    o64 = orc.sttran_forward(entry, sd, dtype=np.float64)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        assert o64[k].dtype == np.float64
        np.testing.assert_allclose(o64[k], g[k], atol=TOL, rtol=0)


def test_single_frame_returns_encoder_output(weights):
    """lib/transformer_wk.py:187-188: with one frame there is no temporal window."""
    sd = _weights(weights, 7)
    entry = syn.make_entry(5, [3])
    st = {}
    orc.sttran_forward(entry, sd, stages=st)
    np.testing.assert_array_equal(st["global_output"], st["local_output"])


def test_unsorted_im_idx_rejected():
    with pytest.raises(ValueError):
        orc.frame_counts_from_im_idx(np.array([0, 1, 0], dtype=np.float32))


@pytest.mark.parametrize("name", ["dsgdetr_4x3", "dsgdetr_ragged", "dsgdetr_16x12", "dsgdetr_shuffled_boxes",
                                  "dsgdetr_empty_frames"])
def test_dsg_detr_oracle_matches_reference(name, golden_dir):
    """DSG-DETR (lib/dsg_detr.py, sgdet branch) restatement vs the imported reference."""
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    sd = syn.make_dsg_detr_state_dict(int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode="sgdet", im_idx_dtype=np.int64)
    if "box_shuffle_seed" in g.files:
        entry = syn.shuffle_boxes(entry, int(g["box_shuffle_seed"]))
    st = {}
    out = orc.dsg_detr_forward(entry, sd, stages=st)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution", "distribution"):
        np.testing.assert_allclose(out[k], g[k], atol=5e-5, rtol=0, err_msg=k)
    if "local_output" in g.files:
        np.testing.assert_allclose(st["local_output"], g["local_output"], atol=5e-5, rtol=0)
    else:
        np.testing.assert_allclose(st["local_output"][:4], g["local_output_head"], atol=5e-5, rtol=0)


# ---- the second restatement (oracle/sttran_torch.py: plain torch on oneDNN, bench.py's other cpu_baseline sample) -------------
@pytest.mark.parametrize("name", CASES)
def test_torch_restatement_matches_reference(name, golden_dir, weights):
    from oracle import sttran_torch as ort
    g = np.load(os.path.join(golden_dir, f"sttran_{name}.npz"))
    mode = "sgdet" if "distribution" in g.files else "predcls"
    sd = _weights(weights, int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode=mode,
                           im_idx_dtype=np.int64 if mode == "sgdet" else np.float32)
    out = ort.sttran_forward(entry, sd, mode=mode)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution") + (("distribution",) if mode == "sgdet" else ()):
        np.testing.assert_allclose(out[k], g[k], atol=TOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("name", ["dsgdetr_4x3", "dsgdetr_ragged", "dsgdetr_shuffled_boxes", "dsgdetr_empty_frames"])
def test_torch_restatement_dsg_detr_matches_reference(name, golden_dir):
    from oracle import sttran_torch as ort
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    sd = syn.make_dsg_detr_state_dict(int(g["weight_seed"]))
    entry = syn.make_entry(int(g["entry_seed"]), g["pairs_per_frame"].tolist(), mode="sgdet", im_idx_dtype=np.int64)
    if "box_shuffle_seed" in g.files:
        entry = syn.shuffle_boxes(entry, int(g["box_shuffle_seed"]))
    out = ort.dsg_detr_forward(entry, sd)
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution", "distribution"):
        np.testing.assert_allclose(out[k], g[k], atol=5e-5, rtol=0, err_msg=k)
