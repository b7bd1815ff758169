"""world_size-2 gloo tests of the clip sharding + prediction all-gather (the N>1 path of bench.py
and of a sharded evaluation run).  CPU only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nl_vsgg_amd.lib.distributed import PredictionGatherer, all_gather_predictions, assign_clips, pack_predictions


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


def _rows(i, n, step=0):
    return torch.full((n, 26), float(i) + 100.0 * step) + torch.arange(n).float()[:, None] * 1e-3


def _worker_pipelined(rank, world, port, pairs, steps, q):
    """the form bench.py --gpus N runs: fixed capacities, a ring of two buffer sets, one submit per step, results
    read only afterwards (and while later gathers are already in flight)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        owner = assign_clips(pairs, world)
        mine = [i for i, o in enumerate(owner) if o == rank]
        rows_cap = max(sum(pairs[i] for i, o in enumerate(owner) if o == r) for r in range(world))
        clips_cap = max(sum(1 for o in owner if o == r) for r in range(world))
        g = PredictionGatherer(rows_cap, clips_cap, cols=26, depth=2)
        ok = True
        tickets = []
        for step in range(steps):
            buf = g.payload()                                  # write the rows straight into the send buffer
            n = sum(pairs[i] for i in mine)
            if n:
                buf[:n] = torch.cat([_rows(i, pairs[i], step) for i in mine])
            tickets.append(g.submit(buf[:n], mine, [pairs[i] for i in mine]))
            if step >= 1:                                      # read the previous step while this one is in flight
                got = g.result(tickets[step - 1])
                ok = ok and sorted(got) == list(range(len(pairs)))
                for i, t in got.items():
                    ok = ok and torch.equal(t, _rows(i, pairs[i], step - 1))
        g.wait_all()
        got = g.result(tickets[-1])
        for i, t in got.items():
            ok = ok and torch.equal(t, _rows(i, pairs[i], steps - 1))
        try:
            g.result(tickets[0])                               # its buffers were reused long ago
            ok = False
        except ValueError:
            pass
        # the one-shot form with both capacities given (no size exchange)
        local = torch.cat([_rows(i, pairs[i]) for i in mine]) if mine else torch.zeros((0, 26))
        got = all_gather_predictions(local, mine, [pairs[i] for i in mine], rows_cap=rows_cap, clips_cap=clips_cap)
        ok = ok and all(torch.equal(got[i], _rows(i, pairs[i])) for i in range(len(pairs)))
        # pack_predictions into the send buffer
        pred = {"attention_distribution": torch.rand(3, 3), "spatial_distribution": torch.rand(3, 6),
                "contacting_distribution": torch.rand(3, 17)}
        view = pack_predictions(pred, out=torch.zeros(5, 26))
        ok = ok and torch.equal(view, pack_predictions(pred)) and view.shape == (3, 26)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pairs", [[5, 1, 7, 3, 3], [4], [2, 2, 2, 2, 6, 1]])
def test_pipelined_gatherer_world2(pairs):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, pairs, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_gatherer_rejects_overflow():
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = PredictionGatherer(rows_cap=4, clips_cap=2, cols=26)
        # capacities are rank-local: an overflowing rank still takes part in the collective and the error is raised
        # where the result is read (by every rank), not before the all-gather
        with pytest.raises(ValueError):
            g.result(g.submit(torch.zeros(5, 26), [0], [5]))
        with pytest.raises(ValueError):
            g.result(g.submit(torch.zeros(3, 26), [0, 1, 2], [1, 1, 1]))
        with pytest.raises(ValueError):
            g.submit(torch.zeros(3, 26), [0], [2])                   # inconsistent arguments: a caller bug, raised at once
        got = g.result(g.submit(torch.ones(3, 26), [7], [3]))
        assert list(got) == [7] and torch.equal(got[7], torch.ones(3, 26))
    finally:
        dist.destroy_process_group()


def _worker_gathered(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank knows every rank's layout (a deterministic assignment): `gathered` hands out the raw buffers without
        # reading the clip records back -- what bench.py's strong-scaling loop uses on rank 0
        rows_of = [3, 5]
        g = PredictionGatherer(rows_cap=6, clips_cap=1, cols=26, depth=2)
        ok = True
        for step in range(3):
            t = g.submit(_rows(rank, rows_of[rank], step), [rank], [rows_of[rank]])
            buf, rec = g.gathered(t)
            ok = ok and tuple(buf.shape) == (world, 6, 26) and tuple(rec.shape) == (world, 1, 2)
            for r in range(world):
                ok = ok and torch.equal(buf[r, :rows_of[r]], _rows(r, rows_of[r], step)) and rec[r, 0].tolist() == [r, rows_of[r]]
        try:
            g.gathered(0)                                  # two submits ago: its buffer set has been reused
            ok = False
        except ValueError:
            pass
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gatherer_gathered_hands_out_raw_buffers_world2():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gathered, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _worker_overflow(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PredictionGatherer(rows_cap=4, clips_cap=2, cols=26)
        n = 6 if rank == 1 else 2                                     # rank 1 alone exceeds rows_cap
        t = g.submit(torch.ones(n, 26), [rank], [n])                  # must not raise before the collective (rank 0 would hang)
        try:
            g.result(t)
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "rank 1" in str(e)))
        # the gatherer stays usable
        got = g.result(g.submit(torch.full((2, 26), float(rank)), [rank], [2]))
        q.put((rank, sorted(got) == [0, 1] and all(torch.equal(got[r], torch.full((2, 26), float(r))) for r in (0, 1))))
    finally:
        dist.destroy_process_group()


def test_gatherer_overflow_on_one_rank_is_seen_by_all_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overflow, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get() for _ in range(4)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok is True for _, ok in res), res


def test_assign_clips_balances_and_is_deterministic():
    costs = [100, 90, 10, 10, 10, 10, 50, 40]
    owner = assign_clips(costs, 4)
    load = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(4)]
    assert max(load) <= 100 and sum(load) == sum(costs)
    assert owner == assign_clips(costs, 4)
    assert assign_clips([], 3) == []
    assert set(assign_clips([1] * 8, 8)) == set(range(8))


# ---- world 4: unequal packs per rank, a rank without clips, sharded scoring merged by ONE all-reduce of tallies -----------
OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]; SPA = [f"spa{i}" for i in range(6)]; CON = [f"con{i}" for i in range(17)]


def _scored_clip(i, frames):
    """clip i of the test set: a synthetic predcls entry with random 'predictions' + its ground truth (a deterministic
    function of i, so every rank can build any clip)"""
    import numpy as np
    from nl_vsgg_amd.lib import synthetic as syn
    rng = np.random.default_rng([77, i])
    e = syn.make_entry(1000 + i, rng.integers(1, 5, frames).tolist(), mode="predcls")
    P = e["pair_idx"].shape[0]
    pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
    pred["attention_distribution"] = rng.standard_normal((P, 3)).astype(np.float32)
    pred["spatial_distribution"] = rng.uniform(0, 1, (P, 6)).astype(np.float32)
    pred["contacting_distribution"] = rng.uniform(0, 1, (P, 17)).astype(np.float32)
    return pred, syn.make_gt_annotation_hard(500 + i, e, jitter=6.0)


def _evaluator():
    from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator
    ev = SceneGraphEvaluator(mode="predcls", AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON,
                             AG_attention_predicates=ATT, AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)
    ev.register_container()
    return ev


FRAMES4 = [9, 2, 7, 3, 3, 5, 2]          # 7 clips over 4 ranks with costs that leave the ranks 1..3 clips; + rank 3 emptied below


FRAMES8 = [9, 2, 7, 3, 3, 5, 2, 8, 4, 6, 2, 2, 5, 3, 7, 4, 2, 6, 3]     # 19 clips over 8 ranks (7 owners + one rank emptied)


def _worker_world4(rank, world, port, q, FRAMES4=FRAMES4):
    import numpy as np
    from nl_vsgg_amd.lib.distributed import all_reduce_recall
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = [float(f) * f for f in FRAMES4]
        owner = assign_clips(costs, world - 1)             # ranks 0..2 own clips, rank 3 owns NONE (an empty rank)
        lists = [[i for i, o in enumerate(owner) if o == r] for r in range(world)]
        pack = 2                                           # clips per "forward": unequal numbers of packs per rank
        packs = [[l[j:j + pack] for j in range(0, len(l), pack)] for l in lists]
        rounds = max(len(p) for p in packs)
        clips = {i: _scored_clip(i, FRAMES4[i]) for i in lists[rank]}
        rows_of = lambda i: int(clips[i][0]["pair_idx"].shape[0])
        pairs_all = [int(_scored_clip(i, f)[0]["pair_idx"].shape[0]) for i, f in enumerate(FRAMES4)]
        rows_cap = max(sum(pairs_all[i] for i in pk) for pq in packs for pk in pq)
        g = PredictionGatherer(rows_cap, pack, cols=26, depth=2)
        ev = _evaluator()
        ok = True
        for r_ in range(rounds):
            ids = packs[rank][r_] if r_ < len(packs[rank]) else []          # some ranks have run out of packs: empty submit
            for i in ids:
                ev.evaluate_scene_graph(clips[i][1], clips[i][0])           # every rank scores ITS OWN clips
            rows = [torch.from_numpy(np.concatenate([clips[i][0][k] for k in ("attention_distribution", "spatial_distribution",
                                                                             "contacting_distribution")], 1)) for i in ids]
            local = torch.cat(rows) if rows else g.payload()[:0]
            t = g.submit(local, ids, [rows_of(i) for i in ids])
            buf, rec = g.gathered(t)                                        # the predictions of every rank, on every rank
            for q_ in range(world):
                want = packs[q_][r_] if r_ < len(packs[q_]) else []
                ok = ok and [int(c) for c, _ in rec[q_].tolist() if c >= 0] == want
                off = 0
                for i in want:
                    p, _ = _scored_clip(i, FRAMES4[i])
                    ok = ok and torch.equal(buf[q_, off:off + pairs_all[i], :3], torch.from_numpy(p["attention_distribution"]))
                    off += pairs_all[i]
        g.raise_if_overflowed()
        merged = all_reduce_recall(ev)                                      # one all-reduce of (sum, count) tallies
        q.put((rank, bool(ok), merged, [len(p) for p in packs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,FRAMES4", [(4, FRAMES4), (8, FRAMES8)])
def test_world4_unequal_packs_empty_rank_and_merged_recall(world, FRAMES4):
    """world 4, and world 8 = the rank count of the driver's one SCALE shot (more than 4 ranks of the
    gatherer / LPT / tally code had never run)"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world4, args=(r, world, port, q, FRAMES4)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert len(res) == world and all(ok for _, ok, _, _ in res), res
    npacks = res[0][3]
    assert npacks[world - 1] == 0 and len(set(npacks[:world - 1])) > 1, npacks     # an empty rank, unequal pack counts elsewhere
    # the merged table == one evaluator over all clips in one process (the reference's loop), on every rank
    ev = _evaluator()
    for i, f in enumerate(FRAMES4):
        pred, gt = _scored_clip(i, f)
        ev.evaluate_scene_graph(gt, pred)
    ev.calculate_mean_recall()
    want = ev.summary()
    for _, _, merged, _ in res:
        for t in want:
            for k in want[t]:
                assert merged[t][k] == pytest.approx(want[t][k], abs=1e-12), (t, k)
    # and the single-process identity: summary_from_partial_sums(partial_sums()) == summary()
    same = ev.summary_from_partial_sums(ev.partial_sums())
    assert all(same[t][k] == pytest.approx(want[t][k], abs=1e-12) for t in want for k in want[t])


def test_gathered_counts_overflow_records():
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = PredictionGatherer(rows_cap=4, clips_cap=2, cols=26)
        g.gathered(g.submit(torch.ones(3, 26), [0], [3]))
        g.raise_if_overflowed()                                             # nothing overflowed: no error
        g.gathered(g.submit(torch.zeros(5, 26), [0], [5]))                  # does not fit: the raw rows are not predictions
        with pytest.raises(ValueError):
            g.raise_if_overflowed()
        g.raise_if_overflowed()                                             # reported once
        # asking twice for the same ticket counts its OVERFLOW record once ...
        t = g.submit(torch.zeros(5, 26), [0], [5])
        g.gathered(t); g.gathered(t)
        assert int(g._overflow_seen) == 1
        with pytest.raises(ValueError):
            g.raise_if_overflowed()
        # ... and a loop that only SUBMITS (bench.py's timed pass never calls gathered() / result()) still hears of it:
        # the record is counted on the first wait for the buffer set -- ring reuse or wait_all
        for _ in range(3):
            g.submit(torch.zeros(5, 26), [0], [5])
        g.wait_all()
        assert int(g._overflow_seen) == 3
        with pytest.raises(ValueError):
            g.raise_if_overflowed()
        # all_reduce_recall on a 1-rank group runs the collective (identity) and returns the evaluator's own summary
    finally:
        dist.destroy_process_group()
