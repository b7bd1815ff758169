"""world_size-2 gloo tests of the clip sharding + prediction all-gather (the N>1 path of bench.py
and of a sharded evaluation run).  CPU only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nl_vsgg_amd.lib.distributed import all_gather_predictions, assign_clips


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        owner = assign_clips([p * 3 for p in pairs], world)
        mine = [i for i, o in enumerate(owner) if o == rank]
        # a clip's "prediction" is a deterministic function of its id: any rank can check any clip
        rows = [torch.full((pairs[i], 26), float(i)) + torch.arange(pairs[i]).float()[:, None] * 1e-3 for i in mine]
        local = torch.cat(rows) if rows else torch.zeros((0, 26))
        got = all_gather_predictions(local, mine, [pairs[i] for i in mine])
        ok = sorted(got) == list(range(len(pairs)))
        for i, t in got.items():
            exp = torch.full((pairs[i], 26), float(i)) + torch.arange(pairs[i]).float()[:, None] * 1e-3
            ok = ok and t.shape == exp.shape and torch.equal(t, exp)
        q.put((rank, ok, owner))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pairs", [[5, 1, 7, 3, 3], [4], [2, 2, 2, 2]])
def test_all_gather_predictions_world2(pairs):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]                 # both ranks derived the same clip -> rank map


def test_assign_clips_balances_and_is_deterministic():
    costs = [100, 90, 10, 10, 10, 10, 50, 40]
    owner = assign_clips(costs, 4)
    load = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(4)]
    assert max(load) <= 100 and sum(load) == sum(costs)
    assert owner == assign_clips(costs, 4)
    assert assign_clips([], 3) == []
    assert set(assign_clips([1] * 8, 8)) == set(range(8))
