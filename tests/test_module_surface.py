"""The shim is an `nn.Module` like the reference's `STTran` (lib/sttran.py:316): the call sequences of
lib/ults/init_teacher_model.py:15-29 and tools/test_STTran.py:38-52 run against it verbatim.
CPU part: everything up to (not including) the first forward -- no handle is created before that; GPU part: the same
sequence followed by forwards, a `state_dict()` round trip into a second model and a forward hook."""
import os
import types

import numpy as np
import pytest
import torch

from nl_vsgg_amd.lib import synthetic as syn
from nl_vsgg_amd.lib.sttran import STTran

CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]


def _dataset():
    # what the two call sites read off the dataset object
    return types.SimpleNamespace(attention_relationships=["a"] * 3, spatial_relationships=["s"] * 6,
                                 contacting_relationships=["c"] * 17, object_classes=CLASSES, object_classes4gt=CLASSES)


def _init_teacher_model(model_conf, AG_dataset_train, gpu_device, conf, ckpt):
    """lib/ults/init_teacher_model.py:15-29, statement for statement (the checkpoint comes in instead of torch.load)"""
    t_model = STTran(mode=model_conf.mode,
                     attention_class_num=len(AG_dataset_train.attention_relationships),
                     spatial_class_num=len(AG_dataset_train.spatial_relationships),
                     contact_class_num=len(AG_dataset_train.contacting_relationships),
                     obj_classes=AG_dataset_train.object_classes,
                     enc_layer_num=model_conf.enc_layer,
                     dec_layer_num=model_conf.dec_layer,
                     transformer_mode=model_conf.transformer_mode,
                     is_wks=model_conf.is_wks,
                     feat_dim=model_conf.feat_dim,
                     conf=conf
                     ).to(device=gpu_device)
    t_model.load_state_dict(ckpt['state_dict'], strict=False)
    return t_model


def _test_sttran_setup(conf, AG_dataset_test, gpu_device, ckpt):
    """tools/test_STTran.py:38-52"""
    model = STTran(mode=conf.mode,
                   attention_class_num=len(AG_dataset_test.attention_relationships),
                   spatial_class_num=len(AG_dataset_test.spatial_relationships),
                   contact_class_num=len(AG_dataset_test.contacting_relationships),
                   obj_classes=AG_dataset_test.object_classes4gt,
                   enc_layer_num=conf.enc_layer,
                   dec_layer_num=conf.dec_layer,
                   transformer_mode=conf.transformer_mode,
                   is_wks=conf.is_wks,
                   feat_dim=conf.feat_dim,
                   conf=conf
                   ).to(device=gpu_device)
    model.eval()
    model.load_state_dict(ckpt['state_dict'], strict=False)
    return model


CONF = types.SimpleNamespace(mode="predcls", enc_layer=1, dec_layer=3, transformer_mode="wk", is_wks=True, feat_dim=2048)


def _ckpt():
    return {"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}}


def test_reference_call_sequences_run_verbatim_without_a_gpu(monkeypatch):
    if not torch.cuda.is_available():
        monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)       # torch.device("cuda") has no index
    ckpt = _ckpt()
    for build in (lambda: _init_teacher_model(CONF, _dataset(), torch.device("cuda"), CONF, ckpt),
                  lambda: _test_sttran_setup(CONF, _dataset(), torch.device("cuda"), ckpt)):
        m = build()
        assert isinstance(m, torch.nn.Module) and m.training is False and m.eval() is m
        # empty-but-typed parameter iterators: `for p in model.parameters(): p.requires_grad = False` is a no-op loop
        assert list(m.parameters()) == [] and list(m.named_parameters()) == [] and m.requires_grad_(False) is m
        sd = m.state_dict()
        assert list(sd.keys()) == list(ckpt["state_dict"].keys())
        assert all(sd[k] is ckpt["state_dict"][k] or torch.equal(sd[k], ckpt["state_dict"][k]) for k in sd)
        assert m.state_dict(prefix="teacher.").keys() == {"teacher." + k for k in sd}
        # lib/pytorch_misc.py:16-style use: `own_state = network.state_dict()` then per-key copy
        own = m.state_dict()
        assert own["vr_fc.weight"].shape == (512, 12544)
        with pytest.raises(NotImplementedError):
            m.train()
        with pytest.raises(RuntimeError):
            m.to("cpu")
        assert m.to(torch.float32) is m                                    # a dtype-only `.to` changes nothing
        assert m._handle is None and m.sync_check() is None                # no forward yet: no handle, nothing to wait for


@pytest.mark.gpu
def test_module_surface_on_the_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    gpu_device = torch.device("cuda")
    ckpt = torch.load if False else _ckpt()
    model = _test_sttran_setup(CONF, _dataset(), gpu_device, ckpt)
    teacher = _init_teacher_model(CONF, _dataset(), gpu_device, CONF, {"state_dict": model.state_dict()})
    entry_np = syn.make_entry(102, [3, 1, 4, 2, 2])
    mk = lambda: {k: torch.from_numpy(v).cuda() for k, v in entry_np.items() if isinstance(v, np.ndarray)}
    seen = []
    h = model.register_forward_hook(lambda mod, args, out: seen.append(out["attention_distribution"].shape))
    with torch.no_grad():
        pred = model(mk())
        pred_t = teacher(mk())
    h.remove()
    assert seen == [pred["attention_distribution"].shape]
    for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
        assert torch.equal(pred[k], pred_t[k]), k                          # state_dict() round trip: the same weights
    assert model.load_state_dict(ckpt["state_dict"], strict=True).missing_keys == []
    assert os.path.basename(model._lib._name) == "libsttran_hip.so"
