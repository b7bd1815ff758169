#!/usr/bin/env python3
"""bench.py -- frames/sec of the STTran PredCls hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (`STTran.forward`) over one batch of synthetic clips resident
in HBM.  Workload of the line's `value` (default, BASELINE.json configs[1]): clips of 16 frames x 12 boxes
x 2048-d region features (P = 176 pairs per clip), `--clips-per-step` clips per pass (default 64; `batch_sweep`
in the line shows 1 / 16 / 64).  `--workload 64x36` selects configs[3]'s clip shape instead (default 4 clips per
pass).  The default run ALSO measures the 64x36 clip in the same process and reports it under
`workloads["64x36"]` -- north_star's scaling target is quoted on that workload --, the DSG-DETR model of configs[4] under
`workloads["dsgdetr_16x12"]`, a 256-clip sample of the configs[2] stand-in (Action-Genome-test-split-shaped clips,
model + device evaluator) under `workloads["ag_split_shaped"]`, and the one-clip-per-pass rate of the 16x12 clip
(the reference's own batch size) under `one_clip_per_pass`: every BASELINE config has a number in the line.

With N > 1 every rank runs its own clips (whole-clip sharding, weak scaling: per-GPU work is fixed, so
the aggregate grows ~N x unless host glue or the gather contends) and each step's predictions are
all-gathered over RCCL inside the timed region by `lib/distributed.py::PredictionGatherer` (the code the
gloo tests cover): issued asynchronously into a ring of two buffer sets, i.e. under the next forward.
`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its N ranks ITSELF as fresh child
processes -- before this process makes any GPU call -- relays rank 0's JSON line and exits non-zero if a
rank does.  With N > 1 the line's `value` is measured on the 64x36 clip (BASELINE configs[3], the workload
north_star quotes the scaling target on; 16x12 rides in `workloads`), the line lists the device every rank
ran on (`devices`: ordinal + PCI bus id, `distinct_devices`), what rank 0 alone reaches on the same per-GPU
workload while the others idle (`one_rank_alone`), and a STRONG-scaling block (`strong_scaling`: a fixed
clip set sharded with `assign_clips`, one gather per round, every rank scores its own clips with the device
evaluator and the recall tallies are all-reduced; per-rank busy / evaluator time and the LPT imbalance are reported).

Output (rank 0): ONE compact JSON line on stdout -- every contract field, the roofline of the dominant kernel class (the
fp32 MFMA GEMM: algorithmic 2*M*N*K FLOPs / HIP-event time, measured in a second, instrumented run of the same K steps)
with its dominant kernel's row, the CPU baseline (the numpy oracle on this host's cores, bounded sample) and the scalars of
every extra leg; scalars only, < 4 KB (`compact_line`; the driver keeps the last 8 KB of stdout).  The FULL object
(per-kernel-template and per-shape tables, per-rank records, notes) goes to `--detail` / $BENCH_DETAIL (default
bench_detail.json) and to stderr as one `BENCH_DETAIL {...}` line.

Every timed loop alternates TWO batches with different allocations and different per-frame pair counts (same work), so the
library's index-map and chunk-table caches miss on every step, as in a real loop (`config.layout_cache`); `same_batch` is
the cached loop of rounds 1-3 for comparison.  `pcie_inclusive_overlapped` is the rate when every step's inputs start in
pinned host memory (never `value`).

`--profile-only-batch` runs warm-up + the timed steps of the selected workload and nothing else (no
one-clip leg, no second workload, no instrumented leg, no CPU baseline): the form to put under
`rocprofv3 --kernel-trace --stats`, whose per-kernel averages are then per-step averages.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.distributed import PredictionGatherer, pack_predictions  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran, pack_clips  # noqa: E402

CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16X3_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS / 6    # the bf16x3 emulation's roof in fp32-equivalent TFLOP/s (419.5)
# frames, boxes per frame, default clips per step.  A step batches ~10 k pairs (64 clips of 16x12 = 11 264 pairs, 4 clips of
# 64x36 = 8 960): the clip is the BASELINE one, the batch is this framework's (`pack_clips`); measured on one MI355X the
# 16x12 rate goes 15.8 k (1 clip) -> 28.3 k (8) -> 30.2 k (16) -> 31.3 k (32) -> 32.0 k (64) -> 32.5 k (128) frames/s as
# tile quantisation and the stream-K fix-ups amortise; `batch_sweep` in the line re-measures 1 / 16 / default every run.
SHAPES = {"16x12": (16, 12, 64), "64x36": (64, 36, 4)}
# calls in flight in the one-clip-per-pass leg.  3 lanes + the caller's stream = the 4 hardware queues a HIP process gets by
# default (GPU_MAX_HW_QUEUES): measured 2 / 3 / 4 / 6 / 8 lanes = 18.8 / 20.9 / 18.8 / 18.5 / 20.3 k frames/s on a box whose
# serial rate was 14.4 k (tools/experiments/lanes_probe.py --api) -- more lanes than queues share queues again
ONE_CLIP_LANES = {"16x12": 3, "64x36": 2}        # (64x36: 2 lanes 9.88-9.94 k, 4 lanes 9.64-10.0 k, serial 9.26-9.34 k frames/s)
# entries per coalesced group of the one-clip-per-pass leg (`model.coalesce`): the reference's loop body unchanged, K calls
# issued as one by-pointer forward on the next lane
ONE_CLIP_COALESCE = {"16x12": 16, "64x36": 4}
SWEEP_CPS = {"16x12": 16, "64x36": 1}                           # the smaller batch of `batch_sweep` (round 1-2 defaults)


def device_clip(T, N, gen, device, shifted=False):
    """One synthetic clip of T frames x N boxes (1 person + N-1 objects per frame) built on the
    device with the distributions of SURVEY.md 8(d).  `shifted`: the same totals (T frames, T*N boxes, T*(N-1) pairs,
    the same number of window tokens) with ONE object moved from frame T//4 to frame T//2 -- another per-frame pair-count
    vector, i.e. another layout for the library's index-map cache, at the same work."""
    counts = np.full(T, N - 1, dtype=np.int64)
    if shifted:
        if T < 4 or N < 3:
            raise ValueError("a shifted clip needs >= 4 frames and >= 3 boxes per frame")
        counts[T // 4] -= 1
        counts[T // 2] += 1
    B, P = int(T + counts.sum()), int(counts.sum())
    cd = torch.from_numpy(counts).to(device)
    first = torch.cumsum(cd + 1, 0) - (cd + 1)                                  # the person box of each frame
    fr = torch.arange(T, device=device).repeat_interleave(cd)
    start = torch.cumsum(cd, 0) - cd
    obj = torch.arange(P, device=device) - start[fr] + 1                        # 1 .. pairs of the frame
    labels = torch.randint(2, 37, (B,), device=device, generator=gen)
    labels[first] = 1
    return {
        "features": torch.randn(B, 2048, device=device, generator=gen),
        "union_feat": torch.randn(P, 2048, 7, 7, device=device, generator=gen),
        "spatial_masks": torch.rand(P, 2, 27, 27, device=device, generator=gen) - 0.5,
        "labels": labels,
        "pair_idx": torch.stack([first[fr], first[fr] + obj], dim=1),
        "im_idx": fr.float(),
        "frame_counts": counts.astype(np.int32),
        "num_frames": T,
    }


def cpu_baseline(T, N, sd, budget_s=24.0, model_kind="sttran", threads=None):
    """The numpy oracle (a port of the reference's CPU path, validated against it by the golden
    tests) timed on this host: one clip per run.  BLAS thread counts 8 / 32 / all cores are tried
    (small GEMMs oversubscribe a 256-core host) and the fastest setting is reported with its count;
    `threads` pins the count instead (the second clip shape re-uses the winner of the first)."""
    from oracle import sttran_oracle as orc
    if model_kind == "dsgdetr":
        entry = syn.uniform_clip(11, T, N, mode="sgdet")
        fwd = lambda: orc.dsg_detr_forward(entry, sd)
    else:
        entry = syn.uniform_clip(11, T, N)
        fwd = lambda: orc.sttran_forward(entry, sd)
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # threadpoolctl absent: whatever BLAS defaults to
        threadpool_limits = None
    ncpu = os.cpu_count() or 1
    tries = [min(int(threads), ncpu)] if threads else sorted({min(8, ncpu), min(32, ncpu), ncpu})
    best = None
    for nthr in tries:
        ctx = threadpool_limits(limits=nthr) if threadpool_limits else None
        try:
            t0 = time.perf_counter()
            fwd()                                       # warm-up (BLAS threads, page faults)
            first = time.perf_counter() - t0
            runs = []
            while sum(runs) + first < budget_s / len(tries) and len(runs) < 3:
                t0 = time.perf_counter()
                fwd()
                runs.append(time.perf_counter() - t0)
        finally:
            if ctx is not None:
                ctx.restore_original_limits()
        med = float(np.median(runs)) if runs else first
        if best is None or med < best[0]:
            best = (med, nthr, max(len(runs), 1))
    med, nthr, nruns = best
    what = "DSG-DETR sgdet" if model_kind == "dsgdetr" else "STTran PredCls"
    return {"value": T / med, "unit": "frames/s", "cores": nthr, "host_cores": ncpu, "kind": "port",
            "sample": f"{nruns} forward(s) of one {T}x{N} clip ({what}), numpy/BLAS fp32 oracle, "
                      + (f"{nthr} BLAS threads" if threads else f"best of 8/32/{ncpu} BLAS threads (cores = the thread count of the best run)")
                      + f", median {med:.3f} s/clip"}


def visible_gpu_count():
    """GPUs this process's children will see, WITHOUT loading a GPU runtime in this process: the KFD topology in sysfs
    (a readable node with SIMDs is a GPU), else the render nodes in /dev/dri; narrowed by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when neither can be read.  ADVISORY: the launcher only
    warns on it -- a rank whose device ordinal does not exist fails in `Env` and the launcher propagates its exit code."""
    n = None
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:                                       # 1. KFD topology: a node with SIMDs is a GPU; a container sees the nodes of
        k = 0                                  #    the whole host but can only READ the properties of its own GPUs
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                k += 1
        n = k
    except (OSError, ValueError):
        pass
    if not n:                                  # 2. the render nodes the process was given
        try:
            n = len([d for d in os.listdir("/dev/dri") if d.startswith("renderD")])
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this process has
    made no GPU call and makes none -- the devices are counted from sysfs, not through the runtime), each with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, let rank 0 print the one JSON line on the inherited stdout, and
    return the first non-zero exit code (the remaining children are then terminated by their own PIDs) or 0."""
    ndev = visible_gpu_count()
    if ndev is not None and ndev < n and "BENCH_FORCE_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but sysfs shows {ndev} GPU(s); starting the ranks anyway (each checks its own device)",
              file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0:
                rc = code
    for pr in live:                                      # a rank failed: stop the others (exact PIDs we started)
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
    return rc


def pci_bus_id(ordinal):
    """PCI bus id of a HIP device ordinal ("0000:c5:00.0")."""
    pr = torch.cuda.get_device_properties(ordinal)
    if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if ctypes.CDLL("libamdhip64.so").hipDeviceGetPCIBusId(buf, 64, int(ordinal)) == 0:
            return buf.value.decode().lower()
    except Exception:
        pass
    return None


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
        table = all_reduce_recall(ev, device=device)           # ONE all-reduce of the (sum, count) tallies
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


class Env:
    """process-wide state shared by the workload runs"""
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run")
        # one process per GPU; BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only so the N>1 code path can be
        # smoke-tested on a single-GPU box (all ranks on device 0, gloo instead of RCCL): tests/test_bench_gpu.py
        self.local = int(os.environ.get("BENCH_FORCE_DEVICE", local))
        ndev = torch.cuda.device_count()
        forced = "BENCH_FORCE_DEVICE" in os.environ
        missing = [r for r in range(self.world) if r >= ndev] if not forced else ([self.rank] if self.local >= ndev else [])
        if missing:
            # A mis-provisioned run (fewer GPUs than ranks; one node: LOCAL_RANK = ordinal) must still leave a parseable
            # record: EVERY rank leaves before the rendezvous (the ranks whose ordinal exists would wait for the others in
            # init_process_group) and rank 0 -- whose ordinal 0 exists whenever any GPU does -- prints a compact line
            # carrying "error"; exit code 2.
            msg = f"rank {missing[0]}: no device {missing[0]} ({ndev} GPU(s) visible, {self.world} rank(s))"
            print(f"bench.py: rank {self.rank}: {msg}", file=sys.stderr)
            if self.rank == 0:
                print(error_line(args, self.world, msg), flush=True)
            raise SystemExit(2)
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        # host threads of this rank: the evaluator's tally, torch's CPU ops and numpy run in this process next to 7 others
        # on an 8-GPU node -- cap torch's intra-op pool so N ranks do not each start one thread per host core
        cores = os.cpu_count() or 1
        self.host_threads = max(1, cores // (2 * self.world)) if self.world > 1 else None
        if self.host_threads:
            torch.set_num_threads(self.host_threads)
        self.dist = None
        if self.world > 1 or getattr(args, "rccl_selftest", False):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1 and "MASTER_PORT" not in os.environ:      # --rccl-selftest without a launcher
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"))    # nccl == RCCL on ROCm
            self.dist = dist

    def devices(self):
        """what every rank ran on, all-gathered: lets the reader check that the N ranks sat on N distinct GPUs"""
        pr = torch.cuda.get_device_properties(self.local)
        mine = {"rank": self.rank, "device": self.local, "pci_bus_id": pci_bus_id(self.local), "name": pr.name,
                "uuid": str(getattr(pr, "uuid", "")) or None, "pid": os.getpid(),
                "backend": self.dist.get_backend() if self.dist else None}
        if self.dist is None:
            return [mine]
        out = [None] * self.world
        self.dist.all_gather_object(out, mine)
        return out

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self, gatherer=None):
        if self.dist is not None:
            if gatherer is not None:
                gatherer.wait_all()
            if self.dist.get_backend() == "nccl":
                self.dist.barrier(device_ids=[self.local])
            else:
                self.dist.barrier()
        torch.cuda.synchronize()


def make_batch(env, model_kind, T, N, cps, seed, shifted=False):
    """cps clips of T x N; `shifted`: clip 0 carries another per-frame pair-count vector (device_clip) at the same totals"""
    device = env.device
    gen = torch.Generator(device=device).manual_seed(seed + env.rank)
    clips = [device_clip(T, N, gen, device, shifted=shifted and i == 0) for i in range(cps)]
    if model_kind == "dsgdetr":                   # sgdet entry: detector boxes, class distribution, scores
        for c in clips:
            B = c["features"].shape[0]
            xy = torch.rand(B, 2, device=device, generator=gen) * 300
            wh = torch.rand(B, 2, device=device, generator=gen) * 150 + 10
            frame_of_box = torch.arange(T, device=device).repeat_interleave(torch.from_numpy(c["frame_counts"] + 1).to(device))
            c["boxes"] = torch.cat([frame_of_box[:, None].float(), xy, xy + wh], 1)
            c["distribution"] = torch.softmax(torch.randn(B, 36, device=device, generator=gen), 1)
            c["scores"] = c["distribution"].max(1).values
            c["im_idx"] = c["im_idx"].long()
    return clips


def by_kernel_tables(entries, forwards):
    """roofline.by_kernel (per kernel template) and roofline.by_shape (per template and problem shape) from the
    library's per-launch-site records: enough to recompute any per-kernel fraction from the bench line alone."""
    fw = max(forwards, 1)
    shape_rows, agg = [], {}
    for e in entries:
        if e["launches"] == 0:
            continue
        row = {"kernel": e["kernel"], "class": e["class"], "M": e["M"], "N": e["N"], "K": e["K"],
               "launches_per_step": e["launches"] / fw, "gflop_per_step": e["flops"] / fw / 1e9,
               "mean_us": 1e3 * e["ms"] / e["launches"]}
        if e["flops"] > 0 and e["ms"] > 0:
            row["tflops"] = e["flops"] / (e["ms"] * 1e-3) / 1e12
        shape_rows.append(row)
        a = agg.setdefault(e["kernel"], {"kernel": e["kernel"], "class": e["class"], "launches": 0, "ms": 0.0, "flops": 0.0})
        a["launches"] += e["launches"]; a["ms"] += e["ms"]; a["flops"] += e["flops"]
    kern_rows = []
    for a in agg.values():
        row = {"kernel": a["kernel"], "class": a["class"], "launches_per_step": a["launches"] / fw,
               "gflop_per_step": a["flops"] / fw / 1e9, "mean_us": 1e3 * a["ms"] / a["launches"],
               "ms_per_step": a["ms"] / fw}
        if a["flops"] > 0 and a["ms"] > 0:
            row["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
            row["frac_of_peak"] = row["tflops"] / FP32_MFMA_PEAK_TFLOPS
        kern_rows.append(row)
    kern_rows.sort(key=lambda r: -r["ms_per_step"])
    shape_rows.sort(key=lambda r: -r["launches_per_step"] * r["mean_us"])
    return kern_rows, shape_rows


def run_workload(env, model, model_kind, workload, cps, steps, warmup, *, graph=False, roofline=True, one_clip=False,
                 pcie=False, repeats=1, alone=False, rotate=True, same_batch=True):
    """Warm-up, EXACTLY `steps` timed steps between barrier + synchronize (max over ranks), then the optional legs."""
    world, device, dist = env.world, env.device, env.dist
    T, N, _ = SHAPES[workload]
    seed = 1234 if workload == "16x12" else 4321
    # TWO batches alternate through every loop below: different allocations (every tensor of every clip) and different
    # per-frame pair counts (clip 0 of batch 1 is "shifted": one object moved between two interior frames -- same frames,
    # boxes, pairs and window tokens, i.e. the same work).  A real loop hands over new tensors with new frame counts on
    # every call (tools/test_STTran.py:81-84), so the library's index-map cache and chunk-table cache MISS on every step:
    # the host-side build_layout and the two staged uploads are inside the timed region.  `same_batch` below re-forwards
    # ONE batch (both caches hit) to show what that costs.
    batches = [make_batch(env, model_kind, T, N, cps, seed + 97 * j, shifted=(j == 1 and rotate)) for j in range(2 if rotate else 1)]
    clips = batches[0]
    P = sum(int(c["pair_idx"].shape[0]) for c in clips)
    model.reserve(P, sum(int(c["features"].shape[0]) for c in clips))
    turn = [0]

    def forward_batch(which=None):
        # The batch is formed HERE, inside the step, from the separate per-clip dicts a producer hands over one at a time
        # (tools/test_STTran.py:81-84): pack_clips(copy=False) passes the clips' own tensors to the library as per-clip
        # pointer tables -- nothing is concatenated, so no copy hides outside the timed region.
        if which is None:
            which = turn[0] % len(batches)
            turn[0] += 1
        b = batches[which]
        return model(pack_clips(b, copy=False)) if cps > 1 else model(dict(b[0]))
    # per-clip predictions of every rank: one fixed-size RCCL all-gather per step (PredictionGatherer)
    gatherer = PredictionGatherer(P, cps, cols=26, device=device, depth=2) if world > 1 else None
    clip_ids = [env.rank * cps + i for i in range(cps)]
    clip_pairs = [int(c["pair_idx"].shape[0]) for c in clips]          # identical for both batches

    def step():
        pred = forward_batch()
        if gatherer is not None:
            gatherer.submit(pack_predictions(pred, out=gatherer.payload()), clip_ids, clip_pairs)
        return pred

    for _ in range(warmup):
        step()
    env.barrier(gatherer)
    run = step
    if graph and world == 1:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            step(); step()                              # warm every lazy path on the capture stream
        torch.cuda.current_stream(device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gpred = step()

        def run():
            g.replay()
            return gpred
        run(); torch.cuda.synchronize()
    # EXACTLY `steps` timed steps between barrier + synchronize, max over ranks -- `repeats` times back to back; the
    # line's value is the MEDIAN repeat (a 20-step region is 0.6 s: one repeat alone moves +-1 % with the box's clocks)
    times = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(steps):
            pred = run()
        env.barrier(gatherer)
        times.append(env.max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(times))
    assert torch.isfinite(pred["attention_distribution"]).all()
    model.sync_check()                  # device-side flags (clamped indices, skipped DSG-DETR sequences) raise here

    frames_per_step = world * cps * T
    res = {
        "value": frames_per_step * steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / steps,
        "steps": steps, "warmup": warmup, "repeats": [frames_per_step * steps / t for t in times],
        "timed_seconds": sum(times),
        "config": {"workload": (f"synthetic {T} frames x {N} boxes x 2048-d region features, STTran PredCls forward "
                                f"(enc 1 / dec 3 layers, d=1936), inputs resident in HBM") if model_kind == "sttran" else
                               (f"synthetic {T} frames x {N} boxes x 2048-d region features, DSG-DETR sgdet forward "
                                f"(1 spatial + 3 temporal encoder layers, d=1936), inputs resident in HBM"),
                   "clips_per_step": cps, "hip_graph": bool(graph and world == 1), "frames_per_clip": T,
                   "batch": (f"formed inside every timed step from {cps} separate per-clip entries, by pointer "
                             f"(pack_clips(copy=False): per-clip pointer tables, no concatenation)") if cps > 1 else "one clip",
                   "boxes_per_frame": N, "pairs_per_step": P,
                   "sharding": f"whole clips, {world} rank(s), one RCCL all-gather of [pairs, 26] prediction rows per step "
                               f"(asynchronous, ring of 2 buffer sets)" if world > 1 else "single GPU",
                   "layout_cache": ("miss every step: two batches of different allocations and different per-frame pair counts "
                                    "alternate, so build_layout and both staged uploads run inside every timed step")
                                   if len(batches) > 1 else "hit (one batch re-forwarded)"},
    }
    if len(batches) > 1 and world == 1 and not graph and same_batch:
        # the loop of rounds 1-3 for comparison: ONE batch re-forwarded, so the index-map and chunk-table caches hit
        for _ in range(2):
            forward_batch(0)
        torch.cuda.synchronize()
        n0 = max(4, min(steps, 20))
        t0 = time.perf_counter()
        for _ in range(n0):
            forward_batch(0)
        torch.cuda.synchronize()
        dt0 = (time.perf_counter() - t0) / n0
        res["same_batch"] = {"value": frames_per_step / dt0, "ms_per_step": 1e3 * dt0, "steps": n0,
                             "layout_cache": "hit", "delta_ms_per_step_vs_value": res["ms_per_step"] - 1e3 * dt0}
    if world > 1:
        # what one gather costs when nothing hides it: back-to-back gathers of the same payload, each waited for
        g2 = PredictionGatherer(P, cps, cols=26, device=device, depth=1)
        rows = pack_predictions(pred)
        for _ in range(3):
            g2.submit(rows, clip_ids, clip_pairs); g2.wait_all()
        env.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            g2.submit(rows, clip_ids, clip_pairs); g2.wait_all()
        torch.cuda.synchronize()
        res["allgather_ms"] = 1e3 * (time.perf_counter() - t0) / 20
        res["allgather_bytes_per_rank"] = P * 26 * 4

        if alone:
            # the same per-GPU workload on rank 0 ALONE while the other ranks wait at the barrier: what one GPU of this
            # node reaches without neighbours (no gather) -- the reference point of the weak-scaling `value`
            dt = None
            if env.rank == 0:
                for _ in range(2):
                    forward_batch()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    forward_batch()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            env.barrier()
            if env.rank == 0:
                res["one_rank_alone"] = {"value": cps * T * steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps,
                                         "note": "rank 0 runs the same per-GPU steps while the other ranks idle; no gather"}

    # ---- the same clip shape, ONE clip per pass (the reference's own batch size; not `value`) -------------
    if one_clip and cps > 1 and world == 1:
        # two clips alternate (clip 0 of each batch: other tensors, other per-frame counts): every call is a new entry, as
        # in the reference's loop.  (a) `serial`: `model(entry)` on the caller's stream, one clip at a time -- rounds 1-3's
        # figure; (b) LANES: the same calls as `model.forward_async(entry)` with ONE_CLIP_LANES lanes in the handle -- call i runs
        # on lane i % lanes' own stream, its result is joined (event wait, no host synchronisation) lanes - 1 calls later, as a pipelined
        # consumer would: one call's launch ramps / prologues / epilogues run under the other calls' MFMAs.
        import collections
        ones = [b[0] for b in batches]
        n1 = 2 * max(steps, 10)

        def loop_serial(n):
            for i in range(n):
                model(dict(ones[i % len(ones)]))

        def loop_lanes(n):
            pending = collections.deque()
            for i in range(n):
                pending.append(model.forward_async(dict(ones[i % len(ones)])))
                if len(pending) == model.lanes:
                    model.join(pending.popleft())
            while pending:
                model.join(pending.popleft())

        def timed(fn, n):
            fn(8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(n)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        dt_serial = timed(loop_serial, n1)
        nlanes = ONE_CLIP_LANES[workload]
        model.lanes = nlanes
        model.reserve(int(ones[0]["pair_idx"].shape[0]) + 8, int(ones[0]["features"].shape[0]) + 8)
        dt1 = timed(loop_lanes, 2 * n1)
        model.sync_check()
        # (c) COALESCED: the same loop body -- `pending.append(model.forward_async(entry))` / `model.join(pred)` -- with
        # `model.coalesce = K`: every K calls are issued as ONE by-pointer forward on the next lane and each entry gets its
        # rows as views; the caller only keeps `model.pipeline_depth` (= lanes x K) entries un-joined instead of `lanes`
        K = ONE_CLIP_COALESCE[workload]
        model.coalesce = K
        model.reserve(K * int(ones[0]["pair_idx"].shape[0]) + 8, K * int(ones[0]["features"].shape[0]) + 8)

        def loop_coalesced(n, hints=True):
            pending = collections.deque()
            for i in range(n):
                e = dict(ones[i % len(ones)])
                if not hints:                              # the reference's entry: no host-side frame counts
                    e.pop("frame_counts"); e.pop("num_frames")
                pending.append(model.forward_async(e))
                if len(pending) == model.pipeline_depth:
                    model.join(pending.popleft())
            while pending:
                model.join(pending.popleft())
        n_co = K * nlanes * max(4, min(steps, 20) // 2)
        loop_coalesced(2 * K * nlanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop_coalesced(n_co)
        torch.cuda.synchronize()
        dt_co = (time.perf_counter() - t0) / n_co
        loop_coalesced(K * nlanes, hints=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop_coalesced(n_co, hints=False)
        torch.cuda.synchronize()
        dt_nh = (time.perf_counter() - t0) / n_co
        model.sync_check()
        model.coalesce = 0
        model.lanes = 1
        res["one_clip_per_pass"] = {"value": T / dt_co, "unit": "frames/s", "ms_per_step": 1e3 * dt_co, "calls": n_co,
                                    "lanes": nlanes, "coalesce": K,
                                    "no_hints": {"value": T / dt_nh, "ms_per_step": 1e3 * dt_nh},
                                    "lanes_only": {"value": T / dt1, "ms_per_step": 1e3 * dt1, "calls": 2 * n1},
                                    "serial": {"value": T / dt_serial, "ms_per_step": 1e3 * dt_serial},
                                    "note": "same clip shape, ONE clip per call: the reference's batch "
                                            "(dataloader/wk_action_genome.py:622-627) and its loop body "
                                            "(tools/test_STTran.py:75-88) as `pending.append(model.forward_async(entry))` / "
                                            f"`model.join(pred)`, a different entry on every call; model.coalesce = {K} issues every "
                                            f"{K} calls as one by-pointer forward on one of {nlanes} lanes; `no_hints` = entries "
                                            "without host-side frame_counts (one im_idx read-back per group); `lanes_only` = "
                                            "coalesce off (round 4's figure); `serial` = one call at a time on the caller's stream"}

    # ---- the headline's batches, two steps in flight on two lanes of the handle (not `value`: steps overlap) ----
    if one_clip and cps > 1 and world == 1:
        import collections
        model.lanes = 2
        model.reserve(P, sum(int(c["features"].shape[0]) for c in clips))

        def loop2(n):
            pending = collections.deque()
            for i in range(n):
                pending.append(model.forward_async(pack_clips(batches[i % len(batches)], copy=False)))
                if len(pending) == 2:
                    model.join(pending.popleft())
            while pending:
                model.join(pending.popleft())
        loop2(4)
        torch.cuda.synchronize()
        n3 = 2 * max(3, min(steps, 12) // 2)
        t0 = time.perf_counter()
        loop2(n3)
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t0) / n3
        model.sync_check()
        model.lanes = 1
        res["two_steps_in_flight"] = {"value": frames_per_step / dt3, "ms_per_step": 1e3 * dt3, "lanes": 2, "steps": n3,
                                      "note": "the same batches with two forwards in flight on two lanes of the handle: the short "
                                              "kernels and tails of one step run under the other step's GEMMs"}

    # ---- the batch size between one clip and the default (the default of rounds 1-2): a few steps, not `value` ---
    if one_clip and world == 1 and cps > SWEEP_CPS[workload] > 1:
        c2 = SWEEP_CPS[workload]
        for _ in range(2):
            for b in batches:
                model(pack_clips(b[:c2], copy=False))
        torch.cuda.synchronize()
        n2 = 2 * max(3, min(steps, 20) // 2)
        t0 = time.perf_counter()
        for i in range(n2):
            model(pack_clips(batches[i % len(batches)][:c2], copy=False))
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / n2
        res["batch_sweep"] = [{"clips_per_step": c2, "value": c2 * T / dt2, "ms_per_step": 1e3 * dt2}]

    # ---- PCIe-inclusive rate (never `value`): inputs start in pinned host memory each step ----------
    if pcie and world == 1:
        batch = pack_clips(clips) if cps > 1 else clips[0]          # one contiguous staging area per tensor
        host = {}
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                host[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                host[k].copy_(v)
        nbytes = sum(v.numel() * v.element_size() for v in host.values())
        if pcie == "full":
            def step_h2d():
                b = dict(batch)
                for k, v in host.items():
                    b[k] = v.to(device, non_blocking=True)
                return model(b)
            for _ in range(2):
                step_h2d()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(steps // 2, 3)):
                step_h2d()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / max(steps // 2, 3)
            res["pcie_inclusive"] = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                     "h2d_bytes_per_step": nbytes,
                                     "note": "serial H2D (pinned) + forward on one stream, no overlap"}
        # the same with the copy of step i+1 on a second stream under the forward of step i (two buffer sets)
        copy_stream, main = torch.cuda.Stream(device), torch.cuda.current_stream(device)
        bufs = [{k: torch.empty_like(batch[k]) for k in host} for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]       # buffer filled
        freed = [torch.cuda.Event() for _ in range(2)]       # forward that read the buffer has finished

        def upload(slot):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[slot])
                for k, v in host.items():
                    bufs[slot][k].copy_(v, non_blocking=True)
                ready[slot].record(copy_stream)

        def pipelined(n):
            for e in freed:
                e.record(main)
            upload(0)
            for i in range(n):
                slot = i & 1
                if i + 1 < n:
                    upload(slot ^ 1)
                main.wait_event(ready[slot])
                b = dict(batch); b.update(bufs[slot])
                model(b)
                freed[slot].record(main)
        pipelined(3)
        torch.cuda.synchronize()
        n_over = max(steps, 6) if pcie == "full" else 6
        t0 = time.perf_counter()
        pipelined(n_over)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_over
        res["pcie_inclusive_overlapped"] = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                            "h2d_bytes_per_step": nbytes, "h2d_gb_per_s": nbytes / dt / 1e9, "steps": n_over,
                                            "note": "inputs start in pinned host memory on every step: H2D of step i+1 on a copy "
                                                    "stream under the forward of step i (never `value`)"}
        del host, bufs, batch

    # ---- roofline of the dominant kernel class: instrumented re-run of the same K steps -------------
    if roofline:
        model.profile(True)
        for _ in range(steps):
            step()
        prof = model.profile_read()
        entries = model.profile_entries()
        model.profile(False, reset=False)
        env.barrier(gatherer)
        gm = prof["gemm"]
        fw = max(prof["forwards"], 1)
        tot_ms = sum(v["ms"] for k, v in prof.items() if isinstance(v, dict))
        ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        # HBM-side traffic of the same kernel class: PMC counters cannot be read from inside this process;
        # tools/pmc_traffic.py turns the two rocprofv3 --pmc passes of `bench.py --profile-only-batch` (FETCH_SIZE x2
        # per the gfx950 correction, WRITE_SIZE) into profiles/*_pmc_traffic_<workload>.json, picked up here (newest
        # round first; the file carries the commit it was taken at).
        traffic, traffic_src, traffic_commit = None, None, None
        pdir = os.path.join(ROOT, "profiles")
        pmc = sorted(p for p in os.listdir(pdir) if p.endswith(f"pmc_traffic_{workload}.json")) if os.path.isdir(pdir) else []
        if pmc and model_kind == "sttran":
            with open(os.path.join(pdir, pmc[-1])) as f:
                pj = json.load(f)
            cls = pj["classes"].get("gemm")
            # per-launch bytes scale with the batch: only a PMC file taken at this run's clips per step applies
            # (files older than round 2's r2_d carry no `clips_per_step`: they were taken at 16 / 1 clips)
            pmc_cps = pj.get("clips_per_step") or {"16x12": 16, "64x36": 1}[workload]
            if cls and pmc_cps == cps:
                traffic, traffic_src, traffic_commit = cls["hbm_bytes_per_launch"], f"profiles/{pmc[-1]}", pj.get("commit")
        by_kernel, by_shape = by_kernel_tables(entries, prof["forwards"])
        dom = next((r for r in by_kernel if r["class"] == "gemm" and "tflops" in r), None)      # sorted by time per step
        res["roofline"] = {
            "kernel": "gemm16_kernel / gemm16c_kernel (v_mfma_f32_16x16x4_f32 tiles 128x176, 128x128, 256x128: nn.Linear launches "
                      "of >= 1 024 rows, conv3x3) + gemm_sk_kernel (32x32x2 tiles: the rest) + their fix-up launches",
            "bound": "mfma", "achieved": ach,
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes per launch (L2 fabric reads x2 + writes)",
            "traffic_source": traffic_src, "traffic_commit": traffic_commit,
            "traffic_measured_in_run": False,      # PMC counters cannot be read in-process: a static, commit-stamped figure
            "algorithmic_bytes_per_launch": gm["bytes"] / max(gm["launches"], 1),
            "launches_per_step": gm["launches"] / fw,
            "avg_launch_us": 1e3 * gm["ms"] / max(gm["launches"], 1),
            "share_of_device_time": gm["ms"] / tot_ms if tot_ms else None,
            "per_class_ms_per_step": {k: v["ms"] / fw for k, v in prof.items() if isinstance(v, dict) and v["launches"]},
            "per_class_tflops": {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in prof.items()
                                 if isinstance(v, dict) and v["ms"] > 0 and v["flops"] > 0},
            "dominant": None if dom is None else {"name": dom["kernel"], "launches_per_step": dom["launches_per_step"],
                                                  "mean_us": dom["mean_us"], "gflop_per_step": dom["gflop_per_step"],
                                                  "tflops": dom["tflops"], "frac": dom["frac_of_peak"],
                                                  "share_of_device_time": dom["ms_per_step"] * fw / tot_ms if tot_ms else None},
            "by_kernel": by_kernel, "by_shape": by_shape,
            "by_kernel_note": "HIP-event time per launch site incl. the stream-K fix-up launch of a GEMM; FLOPs are "
                              "algorithmic 2*M*N*K (unpadded); frac_of_peak vs 157.3 TFLOP/s",
        }
    if model_kind == "sttran":
        # the whole forward against the MFMA ceiling of the REFERENCE's arithmetic (what lib/sttran.py
        # executes per clip, before this implementation's de-duplication / dead-row elimination)
        n, Pc = N - 1, T * (N - 1)
        dec_tok = 2 * n * (T - 1)
        flop_clip = (Pc * (102_238_208 + 45_844_480 + 100_672) + 3 * dec_tok * 45_844_480
                     + Pc * 7_744 * n + 3 * dec_tok * 7_744 * 2 * n)
        eq = res["value"] / world * (flop_clip / T) / 1e12
        res["reference_arithmetic"] = {"gflop_per_frame": flop_clip / T / 1e9, "tflops_equivalent_per_gpu": eq,
                                       "frac_of_fp32_mfma_peak": eq / FP32_MFMA_PEAK_TFLOPS}
    del clips
    torch.cuda.empty_cache()
    return res


COMPACT_LIMIT = 4096


def flush_c_stdio():
    """RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe or a file and would then
    be written at process exit -- BEHIND the JSON line.  Flush it out before the line is printed, so the line stays last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def error_line(args, world, msg):
    """A compact line for a run that could not start (no device for a rank): every contract key, value 0, and `error`."""
    return json.dumps({"metric": "frames/sec (PredCls inference)", "value": 0.0, "unit": "frames/s", "n_gpus": world,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "not run"},
                       "roofline": None, "cpu_baseline": None, "error": msg[:200]})


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
    out["ok"] = bool(ss["gather_verified"] and sorted(got) == [0, 1, 2, 3] and torch.equal(got[2], rows[2 * 2240:3 * 2240]))
    out["seconds"] = time.perf_counter() - t0
    return out


def _r(x, nd=4):
    """floats rounded for the compact line (the detail file keeps full precision)"""
    if isinstance(x, float):
        return round(x, nd) if abs(x) < 1e6 else round(x, 1)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(d):
    """The ONE stdout line: every contract field + scalars of the extra legs, no tables, no prose; < COMPACT_LIMIT bytes.
    `value`, `ms_per_step` and the roofline numbers keep full precision (the driver and the contract tests recompute
    them); everything else is rounded."""
    cfg = d["config"]
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                             "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": f"synthetic {cfg['frames_per_clip']}x{cfg['boxes_per_frame']}x2048-d clips, "
                                 + ("STTran PredCls" if "PredCls" in d["metric"] else "DSG-DETR sgdet") + " forward, inputs in HBM",
                     "clips_per_step": cfg["clips_per_step"], "frames_per_clip": cfg["frames_per_clip"],
                     "boxes_per_frame": cfg["boxes_per_frame"], "pairs_per_step": cfg["pairs_per_step"],
                     "batch": "per-clip pointer tables, formed inside every timed step" if cfg["clips_per_step"] > 1 else "one clip",
                     "layout_cache": cfg["layout_cache"].split(":")[0].split(" (")[0], "hip_graph": cfg["hip_graph"],
                     "sharding": f"whole clips over {d['n_gpus']} rank(s), one all-gather of [pairs,26] rows per step"
                                 if d["n_gpus"] > 1 else "single GPU"}
    out["repeats"] = _r(d["repeats"], 1)
    out["ranks_seen"], out["distinct_devices"] = d["ranks_seen"], d["distinct_devices"]
    if "roofline" in d:
        r = d["roofline"]
        out["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                             "traffic_measured_in_run", "algorithmic_bytes_per_launch", "launches_per_step",
                                             "avg_launch_us", "share_of_device_time")}
        out["roofline"]["kernel"] = "fp32-MFMA GEMM class (gemm16 / gemm16c / gemm_sk kernels + fix-up launches)"
        # FLAT scalars only: the driver's parsed record keeps one level of scalars under `roofline` / `cpu_baseline` and
        # drops nested dicts (BENCH_r04.json lost `dominant{}`), so the dominant kernel's row and the per-class times of
        # one step are spelled out as `dominant_*` / `ms_*` keys; the nested tables live in the detail file
        dom = r.get("dominant") or {}
        for src, dst in (("name", "dominant_kernel"), ("frac", "dominant_frac"), ("tflops", "dominant_tflops"),
                         ("mean_us", "dominant_mean_us"), ("launches_per_step", "dominant_launches_per_step"),
                         ("gflop_per_step", "dominant_gflop_per_step"), ("share_of_device_time", "dominant_share_of_device_time")):
            out["roofline"][dst] = dom.get(src)
        for cls in ("gemm", "union_conv", "mask_conv", "attention", "layernorm", "index"):
            out["roofline"]["ms_" + cls] = _r(float(r["per_class_ms_per_step"].get(cls, 0.0)), 4)
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "host_cores": c["host_cores"],
                               "kind": c["kind"], "sample": c["sample"][:160]}
    for k in ("one_clip_per_pass", "same_batch", "two_steps_in_flight", "pcie_inclusive_overlapped", "one_rank_alone"):
        if k in d:
            out[k] = {"value": _r(d[k]["value"], 1), "ms_per_step": _r(d[k]["ms_per_step"], 4)}
    if "one_clip_per_pass" in d and "serial" in d["one_clip_per_pass"]:
        o = d["one_clip_per_pass"]
        out["one_clip_per_pass"]["lanes"] = o["lanes"]
        out["one_clip_per_pass"]["serial"] = _r(o["serial"]["value"], 1)
        if "coalesce" in o:
            out["one_clip_per_pass"].update(coalesce=o["coalesce"], no_hints=_r(o["no_hints"]["value"], 1),
                                            lanes_only=_r(o["lanes_only"]["value"], 1))
    if "pcie_inclusive_overlapped" in d:
        out["pcie_inclusive_overlapped"]["h2d_gb_per_s"] = _r(d["pcie_inclusive_overlapped"]["h2d_gb_per_s"], 1)
    if "allgather_ms" in d:
        out["allgather_ms"], out["allgather_bytes_per_rank"] = _r(d["allgather_ms"]), d["allgather_bytes_per_rank"]
    if "rccl_selftest" in d:
        st = d["rccl_selftest"]
        out["rccl_selftest"] = {k: _r(st[k]) for k in ("ok", "backend", "allgather_ms", "gather_verified", "seconds") if k in st}
        if "error" in st:
            out["rccl_selftest"]["error"] = str(st["error"])[:120]
    if "batch_sweep" in d:
        out["batch_sweep"] = {str(b["clips_per_step"]): _r(b["value"], 1) for b in d["batch_sweep"]}
    if "reference_arithmetic" in d:
        out["reference_arithmetic_frac"] = _r(d["reference_arithmetic"]["frac_of_fp32_mfma_peak"])
    w = {}
    for name, blk in d.get("workloads", {}).items():
        if "error" in blk:
            w[name] = {"error": blk["error"][:120]}
            continue
        e = {"value": _r(blk["value"], 1)}
        if "ms_per_step" in blk:
            e["ms_per_step"] = _r(blk["ms_per_step"], 3)
        if "roofline" in blk:
            e["roofline_frac"] = _r(blk["roofline"]["frac"])
        if "cpu_baseline" in blk:
            e["cpu_baseline"] = _r(blk["cpu_baseline"]["value"], 1)
        if "two_steps_in_flight" in blk:
            e["two_steps_in_flight"] = _r(blk["two_steps_in_flight"]["value"], 1)
        if "one_clip_per_pass" in blk:
            e["one_clip_per_pass"] = _r(blk["one_clip_per_pass"]["value"], 1)
            if "serial" in blk["one_clip_per_pass"]:
                e["one_clip_serial"] = _r(blk["one_clip_per_pass"]["serial"]["value"], 1)
            if "lanes_only" in blk["one_clip_per_pass"]:
                e["one_clip_lanes_only"] = _r(blk["one_clip_per_pass"]["lanes_only"]["value"], 1)
        if "max_abs_diff_vs_fp32_engine" in blk:
            e["max_abs_diff_vs_fp32_engine"] = blk["max_abs_diff_vs_fp32_engine"]
        if "allgather_ms" in blk:
            e["allgather_ms"] = _r(blk["allgather_ms"])
        w[name] = e
    if w:
        out["workloads"] = w
    ss = {}
    for name, blk in d.get("strong_scaling", {}).items():
        if "error" in blk:
            ss[name] = {"error": blk["error"][:120]}
            continue
        ss[name] = {"value": _r(blk["value"], 1), "seconds": _r(blk["seconds"]), "clips": blk["clips"], "frames": blk["frames"],
                    "ranks": blk["ranks"], "busy_max_s": _r(blk["busy_max_s"]), "eval_max_s": _r(blk["eval_max_s"]),
                    "eval_s_rank0": _r(blk["eval_s_rank0"]), "lpt_imbalance": _r(blk["lpt_imbalance"]),
                    "busy_imbalance": _r(blk["busy_imbalance"]), "gather_verified": blk["gather_verified"],
                    "R@20": blk["recall_with_constraint"].get("20")}
    if ss:
        out["strong_scaling"] = ss
    out["detail"] = "bench_detail.json (also on stderr as BENCH_DETAIL): per-kernel / per-shape tables, per-rank records"
    line = json.dumps(out)
    for k in ("batch_sweep", "reference_arithmetic_frac", "same_batch", "two_steps_in_flight", "detail"):   # never expected; keeps the promise
        if len(line) < COMPACT_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out)
    if len(line) >= COMPACT_LIMIT:
        raise RuntimeError(f"compact bench line is {len(line)} bytes (limit {COMPACT_LIMIT})")
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(SHAPES),
                    help="clip shape of the line's value; default 16x12 (BASELINE configs[1]) on one GPU, 64x36 (configs[3]) on N > 1")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of exactly --steps steps; the value is their median")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling block")
    ap.add_argument("--strong-clips", type=int, default=64, help="clips of 64x36 in the strong-scaling set")
    ap.add_argument("--ag-clips", type=int, default=1737, help="clips of the Action-Genome-split-shaped strong-scaling set")
    ap.add_argument("--clips-per-step", type=int, default=0, help="0 = default for the workload")
    ap.add_argument("--model", default="sttran", choices=["sttran", "dsgdetr"],
                    help="dsgdetr = BASELINE.json configs[4]: lib/dsg_detr.py (sgdet branch) on the same kernels")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and replay it: the forward only "
                         "enqueues on the caller's stream, so it is capturable once the layout is cached")
    ap.add_argument("--pcie", action="store_true",
                    help="the full PCIe leg: serial H2D + forward, and the overlapped form over --steps steps (the default run "
                         "carries a 6-step overlapped leg, `pcie_inclusive_overlapped`)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg of a default run")
    ap.add_argument("--detail", default=os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where rank 0 writes the FULL result object (per-kernel / per-shape tables, per-rank records, notes); "
                         "stdout carries one compact line (< 4 KB) only")
    ap.add_argument("--gemm-engine", default="fp32", choices=["fp32", "bf16x3"],
                    help="bf16x3 = EXPERIMENT: the nn.Linear GEMMs with fp32 emulated on the bf16 matrix pipe (three bf16 planes per "
                         "operand, six cross products, fp32 accumulate); the default line reports it as an extra block only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the second workload (64x36) and the one-clip-per-pass leg of a default run")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1 only: create a one-rank RCCL process group and run the gather / all-reduce / barrier code of the "
                         "N > 1 legs over it; prints {\"rccl_selftest\": {...}} and exits (the default run starts this as a child)")
    ap.add_argument("--no-rccl-selftest", action="store_true")
    ap.add_argument("--profile-only-batch", action="store_true",
                    help="warm-up + timed steps of the selected workload only (for rocprofv3 runs: per-kernel averages "
                         "of the trace are then per-step averages)")
    args = ap.parse_args()
    if args.profile_only_batch:
        args.no_cpu_baseline = args.no_roofline = args.no_extra_workloads = True
        args.repeats = 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, as fresh children, before anything here touches a GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    env = Env(args)
    rank, world, device = env.rank, env.world, env.device
    # N > 1: the line's value is measured on BASELINE configs[3]'s clip (64x36), the workload north_star quotes the
    # scaling target on; N = 1: configs[1] (16x12).  `--workload` overrides.
    if args.workload is None:
        args.workload = "64x36" if world > 1 else "16x12"
    other = "16x12" if args.workload == "64x36" else "64x36"
    T, N, cps_default = SHAPES[args.workload]
    cps = args.clips_per_step or cps_default
    if args.model == "dsgdetr":
        from nl_vsgg_amd.lib.dsg_detr import STTran as DSGDETR
        sd = syn.make_dsg_detr_state_dict(7)
        model = DSGDETR(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                        obj_classes=CLASSES).to(device)
    else:
        sd = syn.make_sttran_state_dict(7)
        model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                       obj_classes=CLASSES, enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True,
                       feat_dim=2048).to(device)
    model.eval()
    model.check_indices = False      # enqueue-only: no per-call synchronisation inside the timed region
    model.strict_inputs = True       # a hidden per-step copy of the inputs would be timed as compute
    model.gemm_engine = args.gemm_engine
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)

    if args.rccl_selftest:
        if world != 1:
            raise SystemExit("--rccl-selftest is the N = 1 leg (N > 1 runs the same code over the real group)")
        try:
            st = rccl_selftest(env, model)
        except Exception as e:
            st = {"ok": False, "error": repr(e)[:300]}
        flush_c_stdio()
        print(json.dumps({"rccl_selftest": st}), flush=True)
        try:
            env.dist.destroy_process_group()
        except Exception:
            pass
        raise SystemExit(0 if st.get("ok") else 1)

    extras = not args.no_extra_workloads
    pcie = "full" if args.pcie else ("overlapped" if extras and world == 1 and not args.no_pcie and not args.graph else False)
    main_res = run_workload(env, model, args.model, args.workload, cps, args.steps, args.warmup, graph=args.graph,
                            roofline=not args.no_roofline, one_clip=extras, pcie=pcie, repeats=args.repeats,
                            alone=extras, rotate=not args.graph, same_batch=not args.profile_only_batch)
    result = {
        "metric": "frames/sec (PredCls inference)" if args.model == "sttran" else "frames/sec (SGDet inference, DSG-DETR)",
        "value": main_res["value"], "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.gemm_engine == "fp32" else "f32 emulated as 3 x bf16 planes (bf16 MFMA, fp32 accumulate) in the "
                                                         "nn.Linear GEMMs; f32 elsewhere",
        "data": "synthetic",
        "config": main_res["config"],
        "repeats": main_res["repeats"], "timed_seconds": main_res["timed_seconds"],
        "value_note": f"median of {len(main_res['repeats'])} back-to-back timed regions of exactly {args.steps} steps each",
    }
    devs = env.devices()
    result["ranks_seen"] = env.dist.get_world_size() if env.dist else 1
    result["devices"] = devs
    result["distinct_devices"] = len({(d["pci_bus_id"], d["uuid"]) if (d["pci_bus_id"] or d["uuid"]) else ("ordinal", d["device"])
                                      for d in devs})
    if world > 1:
        result["scaling_note"] = ("weak scaling: every rank runs the same per-GPU workload on its own clips, so value ~ N x "
                                  "the 1-GPU value by construction unless the host glue or the per-step all-gather contends; "
                                  "compare with `one_rank_alone` of this line or workloads['64x36'] of the --gpus 1 line (the "
                                  "--gpus 1 line's own `value` is the 16x12 clip, BASELINE configs[1]); `strong_scaling` holds "
                                  "the fixed-work legs")
    if "batch_sweep" in main_res:                        # 1 clip / the round 1-2 default / this run's batch, one list
        sweep = []
        if "one_clip_per_pass" in main_res:
            sweep.append({"clips_per_step": 1, "value": main_res["one_clip_per_pass"]["value"],
                          "ms_per_step": main_res["one_clip_per_pass"]["ms_per_step"]})
        sweep += main_res["batch_sweep"]
        sweep.append({"clips_per_step": cps, "value": main_res["value"], "ms_per_step": main_res["ms_per_step"]})
        result["batch_sweep"] = sweep
    for k in ("allgather_ms", "allgather_bytes_per_rank", "one_rank_alone", "one_clip_per_pass", "same_batch", "two_steps_in_flight", "pcie_inclusive",
              "pcie_inclusive_overlapped", "roofline", "reference_arithmetic"):
        if k in main_res:
            result[k] = main_res[k]
    # ---- the other BASELINE clip shape in the same run ----
    if extras and args.model == "sttran":
        steps2 = max(5, min(args.steps, 20))
        w = run_workload(env, model, args.model, other, SHAPES[other][2], steps2, min(args.warmup, 3),
                         roofline=not args.no_roofline, one_clip=(world == 1 and other == "64x36"))
        w.pop("unit", None)
        if "roofline" in w:                              # keep the line readable: per-kernel rows, not per-shape
            w["roofline"].pop("by_shape", None)
        result["workloads"] = {other: w}
    # ---- STRONG scaling (fixed work, any N): 64 clips of 64x36, and the Action-Genome-test-split-shaped set (configs[2]'s
    #      stand-in: frames per clip of ag_test_id.pkl, 1..6 pairs per frame) -- model + gather + device evaluator ----
    if extras and args.model == "sttran" and args.gemm_engine == "fp32" and not args.no_strong:
        result["strong_scaling"] = {}
        with open(os.path.join(ROOT, "tests", "golden", "ag_test_clip_lengths.json")) as f:
            lengths = json.load(f)["frames_per_clip"][:args.ag_clips]
        sets = [("64x36_x64", "64 clips of 64 frames x 36 boxes (BASELINE configs[3]'s clip), STTran PredCls + device Recall@K "
                              "evaluator, 4 clips per forward", [(64, [35] * 64)] * args.strong_clips, 4,
                 lambda sp: float(sp[0]) * float(np.sum(sp[1]))),
                ("ag_split_shaped", f"Action-Genome-test-split-shaped synthetic clips ({len(lengths)} clips, frames per clip from "
                                    "ag_test_id.pkl, 1..6 pairs per frame), STTran PredCls + device Recall@K evaluator, 64 clips "
                                    "per forward, features resident in HBM; the real split's annotations / features are not "
                                    "shipped with the reference", [(int(t), None) for t in lengths], 64,
                 lambda sp: float(sp[0]) * 3.5 * float(sp[0]))]
        for key, name, specs, pack, cost in sets:
            try:
                result["strong_scaling"][key] = strong_scaling(env, model, name, specs, pack, cost)
            except Exception as e:                       # an extra block must never cost the line ...
                if world > 1:
                    raise                                # ... but ranks must not diverge: with N > 1 a failure is fatal
                result["strong_scaling"][key] = {"error": repr(e)}
        if world == 1 and "error" not in result["strong_scaling"]["ag_split_shaped"]:
            result.setdefault("workloads", {})["ag_split_shaped"] = result["strong_scaling"]["ag_split_shaped"]
    # ---- EXPERIMENT block: the same workload with the bf16x3 GEMM engine, and how far its outputs are from the exact
    #      engine's on the same batch (never `value`)
    if extras and world == 1 and args.model == "sttran" and args.workload == "16x12" and args.gemm_engine == "fp32":
        try:
            gen = torch.Generator(device=device).manual_seed(99)
            probe = [device_clip(T, N, gen, device) for _ in range(4)]
            ref = {k: v.clone() for k, v in model(pack_clips(probe, copy=False)).items() if k.endswith("_distribution")}
            model.gemm_engine = "bf16x3"
            got = model(pack_clips(probe, copy=False))
            diff = max(float((got[k] - ref[k]).abs().max()) for k in ref)
            w = run_workload(env, model, args.model, "16x12", cps, max(5, min(args.steps, 20)), min(args.warmup, 3),
                             roofline=not args.no_roofline)
            w.pop("unit", None)
            if "roofline" in w:
                # this engine's roof is the bf16 matrix pipe doing SIX bf16 products per fp32 product: 16 x the fp32-MFMA
                # rate / 6 (MI355X_MICROARCH.md: fp32 MFMA = 1/16 of bf16 MFMA), in fp32-equivalent TFLOP/s
                r = w["roofline"]
                r.pop("by_shape", None)
                r["peak"] = BF16X3_PEAK_TFLOPS
                r["frac"] = r["achieved"] / BF16X3_PEAK_TFLOPS
                r["unit"] = "TFLOP/s (fp32-equivalent: 2*M*N*K per launch)"
                r["kernel"] = ("gemm_x3_kernel + fix-up (v_mfma_f32_32x32x16_bf16, three bf16 planes per operand, six cross "
                               "products) and the launches that stay on the exact engine")
                for row in r.get("by_kernel", []):
                    if "tflops" in row:
                        row["frac_of_peak"] = row["tflops"] / (BF16X3_PEAK_TFLOPS if "x3" in row["kernel"] else FP32_MFMA_PEAK_TFLOPS)
                r["by_kernel_note"] = ("frac_of_peak of gemm_x3_kernel rows vs 419.5 TFLOP/s-equivalent (bf16 dense peak / 6), of the "
                                       "other rows vs the fp32-MFMA peak 157.3")
            w["max_abs_diff_vs_fp32_engine"] = diff
            w["note"] = ("EXPERIMENT, opt-in (model.gemm_engine = 'bf16x3'): nn.Linear GEMMs with M >= 512, the union 1x1 conv and "
                         "the conv3x3 on v_mfma_f32_32x32x16_bf16, each fp32 operand split into three bf16 planes, six cross products, fp32 "
                         "accumulate; error vs fp64 no larger than the exact fp32-MFMA engine's (tests/test_kernels_gpu.py)")
            w.pop("reference_arithmetic", None)          # priced against the fp32 pipe: meaningless for this engine
            result["workloads"]["16x12_bf16x3"] = w
        except Exception as e:
            result["workloads"]["16x12_bf16x3"] = {"error": repr(e)}
        model.gemm_engine = "fp32"
    # ---- the remaining BASELINE configs, driver-witnessed in the same line (single GPU, default run only) ----
    if extras and world == 1 and args.model == "sttran" and args.workload == "16x12":
        del model
        torch.cuda.empty_cache()
        try:                                             # configs[4]: DSG-DETR (lib/dsg_detr.py, sgdet branch) on the same kernels
            from nl_vsgg_amd.lib.dsg_detr import STTran as DSGDETR
            dsd = syn.make_dsg_detr_state_dict(7)
            dm = DSGDETR(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                         obj_classes=CLASSES).to(device)
            dm.eval(); dm.check_indices = False; dm.strict_inputs = True
            dm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in dsd.items()}, strict=False)
            w = run_workload(env, dm, "dsgdetr", "16x12", SHAPES["16x12"][2], max(5, min(args.steps, 20)), min(args.warmup, 3),
                             roofline=not args.no_roofline)
            w.pop("unit", None)
            if "roofline" in w:
                w["roofline"].pop("by_shape", None)
            if rank == 0 and not args.no_cpu_baseline:
                w["cpu_baseline"] = cpu_baseline(16, 12, dsd, model_kind="dsgdetr", budget_s=6.0)
            result["workloads"]["dsgdetr_16x12"] = w
            del dm, dsd
            torch.cuda.empty_cache()
        except Exception as e:                           # an extra block must never cost the line
            result["workloads"]["dsgdetr_16x12"] = {"error": repr(e)}
    if (extras and world == 1 and args.model == "sttran" and args.workload == "16x12" and not args.no_rccl_selftest
            and not args.no_strong and args.gemm_engine == "fp32"):
        # RCCL has no other way to run on a 1-GPU box: a fresh child (never an exec) with a time limit; its verdict rides
        # in the line.  This process's model is gone and its cached blocks were released above.
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-selftest", "--gpus", "1"],
                                capture_output=True, text=True, timeout=240,
                                env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
            last = [l for l in cp.stdout.strip().splitlines() if l.startswith("{")]
            result["rccl_selftest"] = json.loads(last[-1])["rccl_selftest"] if last else {"ok": False, "error": cp.stderr[-300:]}
        except subprocess.TimeoutExpired:
            result["rccl_selftest"] = {"ok": False, "error": "no verdict within 240 s"}
        except Exception as e:
            result["rccl_selftest"] = {"ok": False, "error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "sttran":
        result["cpu_baseline"] = cpu_baseline(T, N, sd)
        if extras and "workloads" in result and other in result["workloads"] and "error" not in result["workloads"][other]:
            # the other clip shape's CPU number, on the thread count that won above (one warm-up + one timed forward)
            result["workloads"][other]["cpu_baseline"] = cpu_baseline(*SHAPES[other][:2], sd, budget_s=14.0,
                                                                      threads=result["cpu_baseline"]["cores"])
    # RCCL writes a version banner through C stdio (block-buffered on a pipe: it would come out at process exit, BEHIND the
    # JSON line) -- every rank flushes it now, ranks other than 0 then close their stdout for good, and rank 0 does the same
    # right behind the line: the line is the LAST thing on the job's stdout whatever the libraries print at teardown.
    flush_c_stdio()
    sys.stdout.flush()
    if rank != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world > 1:
        env.barrier()
    if rank == 0:
        # The FULL object (per-kernel and per-shape tables, per-rank records, notes; ~25 KB) goes to a side file and to
        # stderr; stdout carries ONE compact line of scalars (< 4 KB) -- the driver keeps the last 8 KB of stdout and
        # must find the whole line in it (round 3's 26 KB line could not be parsed).
        try:
            with open(args.detail, "w") as f:
                json.dump(result, f)
        except OSError as e:
            print(f"bench.py: could not write {args.detail}: {e}", file=sys.stderr)
        print("BENCH_DETAIL " + json.dumps(result), file=sys.stderr, flush=True)
        flush_c_stdio()
        print(compact_line(result), flush=True)
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world > 1:
        env.barrier()
        # the line is out; a communicator teardown that does not come back must not keep the launcher (and the driver's
        # clock) waiting: give it 30 s, then leave without it
        import threading
        t = threading.Thread(target=env.dist.destroy_process_group, daemon=True)
        t.start()
        t.join(30.0)
        if t.is_alive():
            print(f"bench.py: rank {rank}: destroy_process_group still running after 30 s, exiting", file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
