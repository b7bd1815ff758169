"""The N > 1 legs: what one all-gather costs, what rank 0 reaches alone, the STRONG-scaling block (fixed clip sets sharded
with `assign_clips`), the one-rank RCCL self-test, and the flat scalars a reader of the driver's SCALE record needs."""
import os
import sys
import time

import numpy as np
import torch

from nl_vsgg_amd.lib.distributed import PredictionGatherer, pack_predictions
from nl_vsgg_amd.lib.sttran import pack_clips

from .common import ROOT


def allgather_cost(w, pred):
    """what one gather costs when nothing hides it: back-to-back gathers of the same payload, each waited for"""
    env = w.env
    g2 = PredictionGatherer(w.P, w.cps, cols=26, device=env.device, depth=1)
    rows = pack_predictions(pred)
    for _ in range(3):
        g2.submit(rows, w.clip_ids, w.clip_pairs); g2.wait_all()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        g2.submit(rows, w.clip_ids, w.clip_pairs); g2.wait_all()
    torch.cuda.synchronize()
    return {"allgather_ms": 1e3 * (time.perf_counter() - t0) / 20, "allgather_bytes_per_rank": w.P * 26 * 4}


def one_rank_alone(w):
    """the same per-GPU workload on rank 0 ALONE while the other ranks wait at the barrier: what one GPU of this node
    reaches without neighbours (no gather) -- the reference point of the weak-scaling `value`.  Returns the block on rank 0,
    None elsewhere."""
    env = w.env
    dt = None
    if env.rank == 0:
        for _ in range(2):
            w.forward_batch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(w.steps):
            w.forward_batch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    env.barrier()
    if env.rank != 0:
        return None
    return {"value": w.cps * w.T * w.steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / w.steps,
            "note": "rank 0 runs the same per-GPU steps while the other ranks idle; no gather"}


def _dist_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class SoloEnv:
    """Rank 0 of an N > 1 job acting as a one-rank job (no collectives): the basis leg of `strong_scaling`."""
    def __init__(self, env):
        self.args, self.rank, self.world, self.device, self.local = env.args, 0, 1, env.device, env.local
        self.host_threads, self.dist = env.host_threads, None

    def max_over_ranks(self, seconds):
        return seconds

    def barrier(self, gatherer=None):
        torch.cuda.synchronize()


def strong_scaling(env, model, name, clip_specs, pack, cost_of):
    """STRONG scaling: a FIXED set of clips (the same whatever N is), sharded over the ranks with `assign_clips`
    (longest-processing-time first on `cost_of`).  Every rank forwards its own clips `pack` per pass AND scores them with
    its own device evaluator (`SceneGraphEvaluator_HIP.evaluate_packed`, one matching kernel per pack); every round's
    `[pairs, 26]` prediction rows are all-gathered to all ranks (`PredictionGatherer`, asynchronous: north_star's
    collective), and at the end ONE all-reduce of the (sum, count) recall tallies (`all_reduce_recall`) gives every rank
    the table of the whole set -- nothing is serialised on rank 0.  Timed: barrier -> merged table on every rank, max over
    ranks, second pass of the process (the first warms the allocator AND verifies the gathered rows of every rank against
    what that rank computed).  Reported per rank: busy time (its forwards), evaluator host time, clips / frames / passes.

    clip_specs[i] = (frames, pairs-per-frame counts or None for the Action Genome range 1..6); a clip is a deterministic
    function of its id and is built by its owner only."""
    if os.path.join(ROOT, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ag_split_bench as ag
    from nl_vsgg_amd.lib.distributed import all_reduce_recall, assign_clips
    from nl_vsgg_amd.lib.evaluation_recall_hip import PackedGroundTruth, SceneGraphEvaluator_HIP
    world, rank, device = env.world, env.rank, env.device
    n = len(clip_specs)
    costs = [cost_of(sp) for sp in clip_specs]
    owner = assign_clips(costs, world)
    order = sorted(range(n), key=lambda i: (-costs[i], i))                     # heaviest first inside every rank
    lists = [[i for i in order if owner[i] == r] for r in range(world)]
    packs = [[l[j:j + pack] for j in range(0, len(l), pack)] for l in lists]
    rounds = max(len(pk) for pk in packs)

    def build(i):
        T, counts = clip_specs[i]
        rng = np.random.default_rng([2024, i])
        gen = torch.Generator(device=device).manual_seed(2024 + i)
        return ag.make_clip(rng, gen, T, device, counts=counts, features=True)

    mine = {i: build(i) for i in lists[rank]}
    ekw = dict(mode="predcls", AG_object_classes=ag.OBJ, AG_all_predicates=ag.ATT + ag.SPA + ag.CON,
               AG_attention_predicates=ag.ATT, AG_spatial_predicates=ag.SPA, AG_contacting_predicates=ag.CON, iou_threshold=0.5)
    # per own pack: the small batch-level tensors the evaluator reads + the pack's ground truth as one device table --
    # data preparation, like the clips themselves
    pack_meta, pack_gt = [], []
    for ids in packs[rank]:
        ents = [{k: mine[i][0][k] for k in ("boxes", "labels", "scores", "pair_idx", "im_idx", "frame_counts", "num_frames")}
                for i in ids]
        meta = pack_clips(ents)
        g = PackedGroundTruth.concat([mine[i][1] for i in ids])
        g.on(device)
        pack_meta.append({k: meta[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores", "num_frames")})
        pack_gt.append(g)
    my_rows = [sum(int(mine[i][0]["pair_idx"].shape[0]) for i in pk) for pk in packs[rank]]
    all_rows = [my_rows]
    coll = env.dist is not None            # N > 1, or the 1-rank RCCL self-test: the same collectives on a group of one
    if coll:
        all_rows = [None] * world
        env.dist.all_gather_object(all_rows, my_rows)
    rows_cap = max(max((max(r, default=1) for r in all_rows), default=1), 1)
    gatherer = PredictionGatherer(rows_cap, pack, cols=26, device=device, depth=2) if coll else None
    model.reserve(max(my_rows, default=1), max((sum(int(mine[i][0]["labels"].shape[0]) for i in pk) for pk in packs[rank]),
                                               default=1))
    stat = {"busy_s": 0.0, "eval_s": 0.0, "gather_mismatch": 0}
    model.lanes = 2
    model.reserve(max(my_rows, default=1), max((sum(int(mine[i][0]["labels"].shape[0]) for i in pk) for pk in packs[rank]),
                                               default=1))

    def one_pass(ev, verify):
        tickets, local_sums, seen = [], [], []
        t_start = time.perf_counter()

        def finish(r_, ids, pred):
            # everything behind a round's forward: join its lane, pack the rows into the gather's send buffer, score the
            # pack on this rank's GPU, issue the round's all-gather (every rank submits once per round, with or without clips)
            rows = None
            if ids:
                model.join(pred)
                rows = pack_predictions(pred, out=gatherer.payload() if gatherer else None)
                t_e = time.perf_counter()
                p = dict(pack_meta[r_])
                for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
                    p[k] = pred[k]
                ev.evaluate_packed(pack_gt[r_], p)             # this rank's own clips, on this rank's GPU
                stat["eval_s"] += time.perf_counter() - t_e
            if gatherer is not None:
                if rows is None:
                    rows = gatherer.payload()[:0]
                if verify:
                    local_sums.append(float(rows.double().sum()))
                tickets.append(gatherer.submit(rows, ids, [int(mine[i][0]["pair_idx"].shape[0]) for i in ids]))
                if verify:                                     # what arrived from every rank (untimed pass only: synchronises)
                    got = gatherer.gathered(tickets[-1])[0]
                    seen.append([float(got[q, :(all_rows[q][r_] if r_ < len(all_rows[q]) else 0)].double().sum())
                                 for q in range(world)])
        # two packs in flight on two lanes of the handle: pack r + 1 is enqueued before pack r is joined, packed and scored,
        # so the short kernels and the tail of one forward run under the next one's GEMMs (+2-4 % at these pack sizes)
        pend = None
        for r_ in range(rounds):
            ids = packs[rank][r_] if r_ < len(packs[rank]) else []
            pred = model.forward_async(pack_clips([mine[i][0] for i in ids], copy=False)) if ids else None
            if pend is not None:
                finish(*pend)
            pend = (r_, ids, pred)
        if pend is not None:
            finish(*pend)
        done = torch.cuda.Event(); done.record()
        done.synchronize()
        stat["busy_s"] = time.perf_counter() - t_start         # this rank's forwards + evaluator launches (enqueue + device)
        t_e = time.perf_counter()
        ev.calculate_mean_recall()                             # flushes the device evaluator: every hit table tallied
        stat["eval_s"] += time.perf_counter() - t_e
        if gatherer is not None:
            gatherer.wait_all()
        # ONE all-reduce of the (sum, count) recall tallies -- over the job's group; a rank acting alone (SoloEnv inside an
        # N > 1 job: torch.distributed IS initialised there) must not enter a collective the other ranks never join
        table = all_reduce_recall(ev, device=device) if (env.dist is not None or not _dist_up()) else \
            ev.summary_from_partial_sums(ev.partial_sums())
        torch.cuda.synchronize()
        if verify and gatherer is not None:
            gatherer.raise_if_overflowed()
            sums = [None] * world
            env.dist.all_gather_object(sums, local_sums)
            for r_ in range(rounds):
                for q in range(world):
                    want = sums[q][r_] if r_ < len(sums[q]) else 0.0
                    if abs(seen[r_][q] - want) > 1e-9 * max(1.0, abs(want)):
                        stat["gather_mismatch"] += 1
        return table

    def fresh():
        e = SceneGraphEvaluator_HIP(**ekw); e.register_container()
        return e
    one_pass(fresh(), True)                                    # untimed: allocator, evaluator kernels, pinned pool; gather verified
    stat["eval_s"] = 0.0
    env.barrier(gatherer)
    ev = fresh()
    t0 = time.perf_counter()
    table = one_pass(ev, False)
    env.barrier(gatherer)
    dt = env.max_over_ranks(time.perf_counter() - t0)
    model.sync_check()
    model.lanes = 1
    frames = sum(sp[0] for sp in clip_specs)
    loads = [sum(costs[i] for i in l) for l in lists]
    per_rank = [{"rank": rank, "clips": len(lists[rank]), "frames": sum(clip_specs[i][0] for i in lists[rank]),
                 "passes": len(packs[rank]), "busy_s": stat["busy_s"], "eval_s": stat["eval_s"],
                 "gather_mismatch": stat["gather_mismatch"]}]
    if coll:
        allr = [None] * world
        env.dist.all_gather_object(allr, per_rank[0])
        per_rank = allr
    mism = sum(p_["gather_mismatch"] for p_ in per_rank)
    if mism:
        raise RuntimeError(f"strong_scaling[{name[:20]}]: {mism} gathered row block(s) differ from what their rank computed")
    busy = [p_["busy_s"] for p_ in per_rank]
    res = {"value": frames / dt, "unit": "frames/s", "seconds": dt, "clips": n, "frames": frames, "ranks": world,
           "clips_per_forward": pack, "rounds": rounds, "per_rank": per_rank,
           "lpt_imbalance": max(loads) / (sum(loads) / world) if sum(loads) else 1.0,
           "busy_imbalance": max(busy) / (sum(busy) / world) if sum(busy) else 1.0,
           "busy_max_s": max(busy), "eval_max_s": max(p_["eval_s"] for p_ in per_rank), "eval_s_rank0": per_rank[0]["eval_s"],
           "gather_verified": gatherer is not None, "host_threads_per_rank": env.host_threads,
           "recall_with_constraint": {str(k): round(float(v), 4) for k, v in table["recall"].items()},
           "config": {"workload": name, "sharding": f"assign_clips (LPT on pairs x frames) over {world} rank(s); every rank scores its own "
                                                    f"clips on its GPU; one all-gather of [pairs, 26] rows per round + one all-reduce of "
                                                    f"the recall tallies"}}
    del mine, pack_meta, pack_gt
    torch.cuda.empty_cache()
    return res



def rccl_selftest(env, model):
    """`bench.py --gpus 1 --rccl-selftest` (run by the default line as a FRESH child process with a time limit, so a
    communicator that does not come up cannot cost the line): a ONE-rank `nccl` (= RCCL) process group in this process and,
    over it, exactly the code the N > 1 legs run -- `strong_scaling` on a small 64x36 set: `PredictionGatherer.submit` under
    the next forward in flight on the handle's lanes, `gathered()` verified against what the rank computed,
    `all_reduce_recall`, `Env.barrier` / `max_over_ranks` -- plus the back-to-back all-gather timing of `run_workload`.
    That loads librccl, creates a communicator and runs its all-gather / all-reduce kernels on the device with the stream
    ordering of lib/distributed.py; what it cannot exercise is the xGMI transport between two GPUs (RCCL refuses two ranks
    on one device: the 2-rank smoke tests stay on gloo)."""
    t0 = time.perf_counter()
    out = {"backend": env.dist.get_backend(), "world": env.world}
    ss = strong_scaling(env, model, "rccl self-test: 8 clips of 64x36, 4 per forward", [(64, [35] * 64)] * 8, 4,
                        lambda sp: float(sp[0]) * float(np.sum(sp[1])))
    out.update(gather_verified=ss["gather_verified"], frames_per_s=ss["value"], rounds=ss["rounds"],
               recall_at_20=ss["recall_with_constraint"].get("20"))
    P = 4 * 64 * 35
    g2 = PredictionGatherer(P, 4, cols=26, device=env.device, depth=1)
    rows = torch.randn(P, 26, device=env.device)
    for _ in range(3):
        g2.submit(rows, [0, 1, 2, 3], [64 * 35] * 4); g2.wait_all()
    env.barrier()
    t1 = time.perf_counter()
    for _ in range(20):
        g2.submit(rows, [0, 1, 2, 3], [64 * 35] * 4); g2.wait_all()
    torch.cuda.synchronize()
    out["allgather_ms"] = 1e3 * (time.perf_counter() - t1) / 20
    got = g2.result(g2.submit(rows, [0, 1, 2, 3], [64 * 35] * 4))
    out["ok"] = bool(out["backend"] == "nccl" and ss["gather_verified"] and sorted(got) == [0, 1, 2, 3] and torch.equal(got[2], rows[2 * 2240:3 * 2240]))
    out["seconds"] = time.perf_counter() - t0
    return out


def strong_sets(args):
    """The two fixed clip sets of the strong-scaling block: (key, name, clip specs, clips per forward, LPT cost)."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ag_test_clip_lengths.json")) as f:
        lengths = json.load(f)["frames_per_clip"][:args.ag_clips]
    return [("64x36_x64", "64 clips of 64 frames x 36 boxes (BASELINE configs[3]'s clip), STTran PredCls + device Recall@K "
                          "evaluator, 4 clips per forward", [(64, [35] * 64)] * args.strong_clips, 4,
             lambda sp: float(sp[0]) * float(np.sum(sp[1]))),
            ("ag_split_shaped", f"Action-Genome-test-split-shaped synthetic clips ({len(lengths)} clips, frames per clip from "
                                "ag_test_id.pkl, 1..6 pairs per frame), STTran PredCls + device Recall@K evaluator, 64 clips "
                                "per forward, features resident in HBM; the real split's annotations / features are not "
                                "shipped with the reference", [(int(t), None) for t in lengths], 64,
             lambda sp: float(sp[0]) * 3.5 * float(sp[0]))]


def strong_block(env, model, args):
    """Both fixed sets on the whole job; with N > 1 additionally the 64x36 set on rank 0 ALONE (`rank0_alone`: the others
    wait), so the record carries its own N = 1 basis for the strong-scaling speed-up."""
    out = {}
    for key, name, specs, pack, cost in strong_sets(args):
        try:
            out[key] = strong_scaling(env, model, name, specs, pack, cost)
        except Exception as e:                       # an extra block must never cost the line ...
            if env.world > 1:
                raise                                # ... but ranks must not diverge: with N > 1 a failure is fatal
            out[key] = {"error": repr(e)}
            continue
        if env.world > 1 and key == "64x36_x64":
            solo = None
            if env.rank == 0:
                s = strong_scaling(SoloEnv(env), model, name + " -- rank 0 alone", specs, pack, cost)
                solo = {"value": s["value"], "seconds": s["seconds"], "recall_with_constraint": s["recall_with_constraint"]}
            env.barrier()
            if solo is not None:
                out[key]["rank0_alone"] = solo
                out[key]["recall_equals_rank0_alone"] = solo["recall_with_constraint"] == out[key]["recall_with_constraint"]
    return out


def scaling_scalars(result, world):
    """FLAT scalars for `config` (where the driver's record keeps them), so that a reader of SCALE_rNN.json with no memory
    of this code never divides a number by one measured on another workload:

    N > 1   one_rank_alone_frames_per_s   rank 0 alone on the per-GPU workload of `value` (same clips, no gather)
            weak_scaling_efficiency       value / (N x one_rank_alone)
            speedup_vs_one_rank           value / one_rank_alone
            allgather_ms, ranks_seen, distinct_devices
            scale_64x36_*                 the same three figures on BASELINE configs[3]'s clip (north_star quotes ">= 6x at
                                          8 GPUs" on it): whole-job frames/s at this N, rank 0 alone, their ratio
            strong_64x36_*                the FIXED 64-clip set: frames/s at this N, on rank 0 alone (the basis), their ratio
    N = 1   scale_reference_64x36_frames_per_s, strong_64x36_frames_per_s: what the N > 1 lines' 64x36 figures divide by."""
    s = {}
    blocks = {result["config"]["frames_per_clip"]: result}
    for name, blk in result.get("workloads", {}).items():
        if isinstance(blk, dict) and "error" not in blk and "config" in blk and "frames_per_clip" in blk["config"]:
            blocks.setdefault(blk["config"]["frames_per_clip"], blk)
    b64 = blocks.get(64)
    strong = (result.get("strong_scaling") or {}).get("64x36_x64") or {}
    if world > 1:
        one = (result.get("one_rank_alone") or {}).get("value")
        if one:
            s["one_rank_alone_frames_per_s"] = one
            s["weak_scaling_efficiency"] = result["value"] / (world * one)
            s["speedup_vs_one_rank"] = result["value"] / one
        if "allgather_ms" in result:
            s["allgather_ms"] = result["allgather_ms"]
        s["ranks_seen"], s["distinct_devices"] = result["ranks_seen"], result["distinct_devices"]
        if b64 is not None:
            s["scale_64x36_frames_per_s"] = b64["value"]
            one64 = (b64.get("one_rank_alone") or {}).get("value")
            if one64:
                s["scale_64x36_one_rank_alone_frames_per_s"] = one64
                s["scale_64x36_speedup_vs_one_rank"] = b64["value"] / one64
        if "value" in strong:
            s["strong_64x36_frames_per_s"] = strong["value"]
            if "rank0_alone" in strong:
                s["strong_64x36_speedup_basis"] = strong["rank0_alone"]["value"]
                s["strong_64x36_speedup"] = strong["value"] / strong["rank0_alone"]["value"]
    else:
        if b64 is not None:
            s["scale_reference_64x36_frames_per_s"] = b64["value"]
        if "value" in strong:
            s["strong_64x36_frames_per_s"] = strong["value"]
    return s
