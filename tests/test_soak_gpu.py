"""Soak: a few hundred forwards over clips whose shape changes every call (index maps rebuilt, staging ring
cycling, packed and single clips mixed), checked now and then against the CPU oracle, and device memory must
stop growing once the workspace has reached its size."""
import numpy as np
import pytest
import torch

from nl_vsgg_amd.lib import synthetic as syn

OUT = ("attention_distribution", "spatial_distribution", "contacting_distribution")


@pytest.mark.gpu
def test_soak_changing_layouts():
    from nl_vsgg_amd.lib.sttran import STTran, pack_clips, unpack_predictions
    from oracle import sttran_oracle as orc
    iters = 240
    sd = syn.make_sttran_state_dict(7)
    m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
               obj_classes=["__background__"] + [f"c{i}" for i in range(36)], enc_layer_num=1, dec_layer_num=3,
               transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    rng = np.random.default_rng(2025)
    pool = []                                          # small clips, reused in random packs
    for i in range(24):
        T = int(rng.integers(1, 9))
        counts = [int(c) for c in rng.integers(0, 7, T)]
        if sum(counts) == 0:
            counts[0] = 2
        e = syn.make_entry(7000 + i, counts)
        pool.append((e, {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v)
                         for k, v in e.items()}))
    free0, worst = None, 0.0
    for it in range(iters):
        k = int(rng.integers(1, 6))
        pick = [int(j) for j in rng.integers(0, len(pool), k)]
        if k == 1:
            pred = [m(dict(pool[pick[0]][1]))]
        else:
            pred = unpack_predictions(m(pack_clips([dict(pool[j][1]) for j in pick])))
        if it % 40 == 0:
            torch.cuda.synchronize()
            for j, p in zip(pick, pred):
                ref = orc.sttran_forward(pool[j][0], sd)
                for key in OUT:
                    worst = max(worst, float(np.abs(p[key].cpu().numpy() - ref[key]).max()))
        if it == iters // 2:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert worst < 1e-3
    assert free0 - free1 < 64 * 2**20, "device memory keeps growing"


@pytest.mark.gpu
def test_soak_lanes_changing_layouts():
    """the same soak on the handle's lanes: 3 calls in flight, layouts and allocations changing on every call, packs and
    single clips mixed, entries and predictions DROPPED by the caller right after submission now and then (the shim must
    keep them alive until the lane is done), every 10th result compared bit for bit with the classic forward"""
    import collections
    from nl_vsgg_amd.lib.sttran import STTran, pack_clips
    sd = syn.make_sttran_state_dict(7)

    def model():
        m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                   obj_classes=["__background__"] + [f"c{i}" for i in range(36)], enc_layer_num=1, dec_layer_num=3,
                   transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
        m.eval(); m.check_indices = False
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
        return m
    m, ref = model(), model()
    m.lanes = 3
    rng = np.random.default_rng(77)

    def fresh_clip(i):                               # new tensors on every call: nothing is ever forwarded twice
        T = int(rng.integers(1, 9))
        counts = [int(c) for c in rng.integers(0, 7, T)]
        if sum(counts) == 0:
            counts[0] = 2
        e = syn.make_entry(9000 + i, counts)
        return {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
    pending = collections.deque()
    checked = 0
    free0 = None
    for it in range(300):
        k = int(rng.integers(1, 4))
        clips = [fresh_clip(10 * it + j) for j in range(k)]
        entry = dict(clips[0]) if k == 1 else pack_clips([dict(c) for c in clips], copy=False)
        pred = m.forward_async(entry)
        if it % 10 == 0:
            want = ref(dict(clips[0]) if k == 1 else pack_clips([dict(c) for c in clips], copy=False))
            pending.append((pred, {key: want[key].clone() for key in OUT}))
        else:
            del pred, entry, clips                   # dropped while the lane may still be reading / writing them
            torch.empty(1 << 22, device="cuda").fill_(float("nan"))     # ... and the allocator is invited to reuse the memory
        while len(pending) > 2:
            p, w = pending.popleft()
            m.join(p)
            got = {key: p[key].clone() for key in OUT}
            torch.cuda.synchronize()
            assert all(torch.equal(got[key], w[key]) for key in OUT), it
            checked += 1
        if it == 150:
            m.sync_check()
            free0 = torch.cuda.mem_get_info()[0]
    m.sync_check()
    for p, w in pending:
        assert all(torch.equal(p[key], w[key]) for key in OUT)
        checked += 1
    assert checked >= 25
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * 2**20, "device memory keeps growing"
