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

One JSON line is printed by rank 0; it carries the roofline of the dominant kernel class (the fp32 MFMA
GEMM: algorithmic 2*M*N*K FLOPs / HIP-event time, measured in a second, instrumented run of the same K
steps, with a per-kernel-template and per-shape breakdown) and a CPU baseline (the numpy oracle on this
host's cores, bounded sample).

`--profile-only-batch` runs warm-up + the timed steps of the selected workload and nothing else (no
one-clip leg, no second workload, no instrumented leg, no CPU baseline): the form to put under
`rocprofv3 --kernel-trace --stats`, whose per-kernel averages are then per-step averages.
"""
import argparse
import json
import os
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
# frames, boxes per frame, default clips per step.  A step batches ~10 k pairs (64 clips of 16x12 = 11 264 pairs, 4 clips of
# 64x36 = 8 960): the clip is the BASELINE one, the batch is this framework's (`pack_clips`); measured on one MI355X the
# 16x12 rate goes 15.8 k (1 clip) -> 28.3 k (8) -> 30.2 k (16) -> 31.3 k (32) -> 32.0 k (64) -> 32.5 k (128) frames/s as
# tile quantisation and the stream-K fix-ups amortise; `batch_sweep` in the line re-measures 1 / 16 / default every run.
SHAPES = {"16x12": (16, 12, 64), "64x36": (64, 36, 4)}
SWEEP_CPS = {"16x12": 16, "64x36": 1}                           # the smaller batch of `batch_sweep` (round 1-2 defaults)


def device_clip(T, N, gen, device):
    """One synthetic clip of T frames x N boxes (1 person + N-1 objects per frame) built on the
    device with the distributions of SURVEY.md 8(d)."""
    B, P = T * N, T * (N - 1)
    fr = torch.arange(T, device=device).repeat_interleave(N - 1)
    obj = torch.arange(1, N, device=device).repeat(T)
    labels = torch.randint(2, 37, (B,), device=device, generator=gen)
    labels[::N] = 1
    return {
        "features": torch.randn(B, 2048, device=device, generator=gen),
        "union_feat": torch.randn(P, 2048, 7, 7, device=device, generator=gen),
        "spatial_masks": torch.rand(P, 2, 27, 27, device=device, generator=gen) - 0.5,
        "labels": labels,
        "pair_idx": torch.stack([fr * N, fr * N + obj], dim=1),
        "im_idx": fr.float(),
        "frame_counts": np.full(T, N - 1, dtype=np.int32),
        "num_frames": T,
    }


def cpu_baseline(T, N, sd, budget_s=24.0):
    """The numpy oracle (a port of the reference's CPU path, validated against it by the golden
    tests) timed on this host: one clip per run.  BLAS thread counts 8 / 32 / all cores are tried
    (small GEMMs oversubscribe a 256-core host) and the fastest setting is reported with its count."""
    from oracle import sttran_oracle as orc
    entry = syn.uniform_clip(11, T, N)
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # threadpoolctl absent: whatever BLAS defaults to
        threadpool_limits = None
    ncpu = os.cpu_count() or 1
    best = None
    for nthr in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
        ctx = threadpool_limits(limits=nthr) if threadpool_limits else None
        try:
            t0 = time.perf_counter()
            orc.sttran_forward(entry, sd)              # warm-up (BLAS threads, page faults)
            first = time.perf_counter() - t0
            runs = []
            while sum(runs) + first < budget_s / 3 and len(runs) < 3:
                t0 = time.perf_counter()
                orc.sttran_forward(entry, sd)
                runs.append(time.perf_counter() - t0)
        finally:
            if ctx is not None:
                ctx.restore_original_limits()
        med = float(np.median(runs)) if runs else first
        if best is None or med < best[0]:
            best = (med, nthr, max(len(runs), 1))
    med, nthr, nruns = best
    return {"value": T / med, "unit": "frames/s", "cores": nthr, "host_cores": ncpu, "kind": "port",
            "sample": f"{nruns} forward(s) of one {T}x{N} clip, numpy/BLAS fp32 oracle, best of 8/32/{ncpu} BLAS "
                      f"threads (cores = the thread count of the best run), median {med:.3f} s/clip"}


def ag_split_sample(device, n_clips):
    """BASELINE configs[2] stand-in on a sample of the split: synthetic clips with the Action Genome test split's
    frames-per-clip (tests/golden/ag_test_clip_lengths.json: the first `n_clips` of its 1 737 clips), 1..6 pairs per frame,
    packed 64 per forward, each pack scored by ONE device-evaluator call (tools/ag_split_bench.py runs all of them)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ag_split_bench as ag
    from nl_vsgg_amd.lib.evaluation_recall_hip import PackedGroundTruth, SceneGraphEvaluator_HIP
    with open(os.path.join(ROOT, "tests", "golden", "ag_test_clip_lengths.json")) as f:
        lengths = json.load(f)["frames_per_clip"][:n_clips]
    model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=ag.OBJ,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(device)
    model.eval(); model.check_indices = False
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}, strict=False)
    ev = SceneGraphEvaluator_HIP(mode="predcls", AG_object_classes=ag.OBJ, AG_all_predicates=ag.ATT + ag.SPA + ag.CON,
                                 AG_attention_predicates=ag.ATT, AG_spatial_predicates=ag.SPA, AG_contacting_predicates=ag.CON,
                                 iou_threshold=0.5)
    ev.register_container()
    rng = np.random.default_rng(2024)
    gen = torch.Generator(device=device).manual_seed(2024)
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    chunk = [ag.make_clip(rng, gen, lengths[i], device) for i in order]
    # ground truth of each pack of 64 clips as one device table (data preparation, untimed like the clips themselves)
    group_gt = {i: PackedGroundTruth.concat([gt for _, gt in chunk[i:i + 64]]) for i in range(0, len(chunk), 64)}
    for g in group_gt.values():
        g.on(device)
    ekw = dict(mode="predcls", AG_object_classes=ag.OBJ, AG_all_predicates=ag.ATT + ag.SPA + ag.CON,
               AG_attention_predicates=ag.ATT, AG_spatial_predicates=ag.SPA, AG_contacting_predicates=ag.CON, iou_threshold=0.5)

    def loop(e):
        for i in range(0, len(chunk), 64):
            e.evaluate_packed(group_gt[i], model(pack_clips([dict(c[0]) for c in chunk[i:i + 64]])))   # one call per pack
        e.calculate_mean_recall()
        torch.cuda.synchronize()
    # one untimed pass first (a throw-away evaluator): the packs' buffers come from the caching allocator afterwards and
    # the evaluator's kernel / pinned pool exist -- the timed pass is the steady state of a long split, not its first second
    w = SceneGraphEvaluator_HIP(**ekw); w.register_container()
    loop(w)
    del w
    t0 = time.perf_counter()
    loop(ev)
    dt = time.perf_counter() - t0
    frames = sum(c[0]["num_frames"] for c in chunk)
    return {"value": frames / dt, "seconds": dt, "clips": len(chunk), "frames": frames,
            "pairs": sum(int(c[0]["pair_idx"].shape[0]) for c in chunk),
            "config": {"workload": "Action-Genome-test-split-shaped synthetic clips (frames per clip from ag_test_id.pkl, 1..6 pairs "
                                   "per frame), STTran PredCls + device Recall@K evaluator, 64 clips per forward, features "
                                   "resident in HBM; the real split's annotations / features are not shipped with the reference"},
            "recall_with_constraint": {str(k): round(float(v), 4) for k, v in ev.summary()["recall"].items()}}


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
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"))    # nccl == RCCL on ROCm
            self.dist = dist

    def barrier(self, gatherer=None):
        if self.world > 1:
            if gatherer is not None:
                gatherer.wait_all()
            if self.dist.get_backend() == "nccl":
                self.dist.barrier(device_ids=[self.local])
            else:
                self.dist.barrier()
        torch.cuda.synchronize()


def make_batch(env, model_kind, T, N, cps, seed):
    device = env.device
    gen = torch.Generator(device=device).manual_seed(seed + env.rank)
    clips = [device_clip(T, N, gen, device) for _ in range(cps)]
    if model_kind == "dsgdetr":                   # sgdet entry: detector boxes, class distribution, scores
        for c in clips:
            B = c["features"].shape[0]
            xy = torch.rand(B, 2, device=device, generator=gen) * 300
            wh = torch.rand(B, 2, device=device, generator=gen) * 150 + 10
            c["boxes"] = torch.cat([torch.arange(T, device=device).repeat_interleave(N)[:, None].float(), xy, xy + wh], 1)
            c["distribution"] = torch.softmax(torch.randn(B, 36, device=device, generator=gen), 1)
            c["scores"] = c["distribution"].max(1).values
            c["im_idx"] = c["im_idx"].long()
    batch = pack_clips(clips) if cps > 1 else clips[0]
    return batch, clips


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
                 pcie=False):
    """Warm-up, EXACTLY `steps` timed steps between barrier + synchronize (max over ranks), then the optional legs."""
    world, device, dist = env.world, env.device, env.dist
    T, N, _ = SHAPES[workload]
    batch, clips = make_batch(env, model_kind, T, N, cps, 1234 if workload == "16x12" else 4321)
    P = int(batch["pair_idx"].shape[0])
    model.reserve(P, int(batch["features"].shape[0]))
    # per-clip predictions of every rank: one fixed-size RCCL all-gather per step (PredictionGatherer)
    gatherer = PredictionGatherer(P, cps, cols=26, device=device, depth=2) if world > 1 else None
    clip_ids = [env.rank * cps + i for i in range(cps)]
    clip_pairs = [T * (N - 1)] * cps

    def step():
        # a fresh dict per call: forward() writes its outputs into the entry (sgdet replaces `distribution`)
        pred = model(dict(batch))
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
    t0 = time.perf_counter()
    for _ in range(steps):
        pred = run()
    env.barrier(gatherer)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(pred["attention_distribution"]).all()

    frames_per_step = world * cps * T
    res = {
        "value": frames_per_step * steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / steps,
        "steps": steps, "warmup": warmup,
        "config": {"workload": (f"synthetic {T} frames x {N} boxes x 2048-d region features, STTran PredCls forward "
                                f"(enc 1 / dec 3 layers, d=1936), inputs resident in HBM") if model_kind == "sttran" else
                               (f"synthetic {T} frames x {N} boxes x 2048-d region features, DSG-DETR sgdet forward "
                                f"(1 spatial + 3 temporal encoder layers, d=1936), inputs resident in HBM"),
                   "clips_per_step": cps, "hip_graph": bool(graph and world == 1), "frames_per_clip": T,
                   "boxes_per_frame": N, "pairs_per_step": P,
                   "sharding": f"whole clips, {world} rank(s), one RCCL all-gather of [pairs, 26] prediction rows per step "
                               f"(asynchronous, ring of 2 buffer sets)" if world > 1 else "single GPU"},
    }
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

    # ---- the same clip shape, ONE clip per pass (the reference's own batch size; not `value`) -------------
    if one_clip and cps > 1 and world == 1:
        one = clips[0]
        for _ in range(3):
            model(dict(one))
        torch.cuda.synchronize()
        n1 = max(2 * steps, 20)
        t0 = time.perf_counter()
        for _ in range(n1):
            model(dict(one))
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / n1
        res["one_clip_per_pass"] = {"value": T / dt1, "unit": "frames/s", "ms_per_step": 1e3 * dt1,
                                    "note": "same clip shape with clips_per_step = 1: the reference's batch "
                                            "(dataloader/wk_action_genome.py:622-627); latency-bound, one clip cannot fill 256 CUs"}
        model(dict(batch))                              # restore the cached layout of the batch

    # ---- the batch size between one clip and the default (the default of rounds 1-2): a few steps, not `value` ---
    if one_clip and world == 1 and cps > SWEEP_CPS[workload] > 1:
        c2 = SWEEP_CPS[workload]
        small = pack_clips([dict(c) for c in clips[:c2]])
        for _ in range(2):
            model(dict(small))
        torch.cuda.synchronize()
        n2 = max(5, min(steps, 20))
        t0 = time.perf_counter()
        for _ in range(n2):
            model(dict(small))
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / n2
        res["batch_sweep"] = [{"clips_per_step": c2, "value": c2 * T / dt2, "ms_per_step": 1e3 * dt2}]
        del small
        model(dict(batch))                              # restore the cached layout of the batch

    # ---- PCIe-inclusive rate (never `value`): inputs start in pinned host memory each step ----------
    if pcie and world == 1:
        host = {k: v.cpu().pin_memory() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        nbytes = sum(v.numel() * v.element_size() for v in host.values())

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
        n_over = max(steps, 6)
        t0 = time.perf_counter()
        pipelined(n_over)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_over
        res["pcie_inclusive_overlapped"] = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                            "note": "H2D of step i+1 on a copy stream under the forward of step i"}

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
        res["roofline"] = {
            "kernel": "gemm_sk_kernel + gemm_fixup_kernel (fp32 MFMA 32x32x2, all nn.Linear / conv3x3 launches)",
            "bound": "mfma", "achieved": ach,
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes per launch (L2 fabric reads x2 + writes)",
            "traffic_source": traffic_src, "traffic_commit": traffic_commit,
            "algorithmic_bytes_per_launch": gm["bytes"] / max(gm["launches"], 1),
            "launches_per_step": gm["launches"] / fw,
            "avg_launch_us": 1e3 * gm["ms"] / max(gm["launches"], 1),
            "share_of_device_time": gm["ms"] / tot_ms if tot_ms else None,
            "per_class_ms_per_step": {k: v["ms"] / fw for k, v in prof.items() if isinstance(v, dict) and v["launches"]},
            "per_class_tflops": {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in prof.items()
                                 if isinstance(v, dict) and v["ms"] > 0 and v["flops"] > 0},
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
    del batch, clips
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="16x12", choices=sorted(SHAPES))
    ap.add_argument("--clips-per-step", type=int, default=0, help="0 = default for the workload")
    ap.add_argument("--model", default="sttran", choices=["sttran", "dsgdetr"],
                    help="dsgdetr = BASELINE.json configs[4]: lib/dsg_detr.py (sgdet branch) on the same kernels")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and replay it: the forward only "
                         "enqueues on the caller's stream, so it is capturable once the layout is cached")
    ap.add_argument("--pcie", action="store_true",
                    help="also report the rate when every step first copies its inputs from pinned host memory")
    ap.add_argument("--gemm-engine", default="fp32", choices=["fp32", "bf16x3"],
                    help="bf16x3 = EXPERIMENT: the nn.Linear GEMMs with fp32 emulated on the bf16 matrix pipe (three bf16 planes per "
                         "operand, six cross products, fp32 accumulate); the default line reports it as an extra block only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the second workload (64x36) and the one-clip-per-pass leg of a default run")
    ap.add_argument("--profile-only-batch", action="store_true",
                    help="warm-up + timed steps of the selected workload only (for rocprofv3 runs: per-kernel averages "
                         "of the trace are then per-step averages)")
    args = ap.parse_args()
    if args.profile_only_batch:
        args.no_cpu_baseline = args.no_roofline = args.no_extra_workloads = True

    env = Env(args)
    rank, world, device = env.rank, env.world, env.device
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

    extras = not args.no_extra_workloads
    main_res = run_workload(env, model, args.model, args.workload, cps, args.steps, args.warmup, graph=args.graph,
                            roofline=not args.no_roofline, one_clip=extras, pcie=args.pcie)
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
    }
    if world > 1:
        result["scaling_note"] = ("weak scaling: every rank runs the same per-GPU workload on its own clips, so value ~ N x "
                                  "the 1-GPU value by construction unless the host glue or the per-step all-gather contends")
    if "batch_sweep" in main_res:                        # 1 clip / the round 1-2 default / this run's batch, one list
        sweep = []
        if "one_clip_per_pass" in main_res:
            sweep.append({"clips_per_step": 1, "value": main_res["one_clip_per_pass"]["value"],
                          "ms_per_step": main_res["one_clip_per_pass"]["ms_per_step"]})
        sweep += main_res["batch_sweep"]
        sweep.append({"clips_per_step": cps, "value": main_res["value"], "ms_per_step": main_res["ms_per_step"]})
        result["batch_sweep"] = sweep
    for k in ("allgather_ms", "allgather_bytes_per_rank", "one_clip_per_pass", "pcie_inclusive", "pcie_inclusive_overlapped",
              "roofline", "reference_arithmetic"):
        if k in main_res:
            result[k] = main_res[k]
    # ---- the other BASELINE workload in the same run (north_star's scaling target is quoted on 64x36) ----
    if extras and args.model == "sttran" and args.workload == "16x12":
        steps2 = max(5, min(args.steps, 20))
        w = run_workload(env, model, args.model, "64x36", SHAPES["64x36"][2], steps2, min(args.warmup, 3),
                         roofline=not args.no_roofline)
        w.pop("unit", None)
        if "roofline" in w:                              # keep the line readable: per-kernel rows, not per-shape
            w["roofline"].pop("by_shape", None)
        result["workloads"] = {"64x36": w}
    # ---- EXPERIMENT block: the same workload with the bf16x3 GEMM engine, and how far its outputs are from the exact
    #      engine's on the same batch (never `value`)
    if extras and world == 1 and args.model == "sttran" and args.workload == "16x12" and args.gemm_engine == "fp32":
        try:
            gen = torch.Generator(device=device).manual_seed(99)
            probe = pack_clips([device_clip(T, N, gen, device) for _ in range(4)])
            ref = {k: v.clone() for k, v in model(dict(probe)).items() if k.endswith("_distribution")}
            model.gemm_engine = "bf16x3"
            got = model(dict(probe))
            diff = max(float((got[k] - ref[k]).abs().max()) for k in ref)
            w = run_workload(env, model, args.model, "16x12", cps, max(5, min(args.steps, 20)), min(args.warmup, 3), roofline=False)
            w.pop("unit", None)
            w["max_abs_diff_vs_fp32_engine"] = diff
            w["note"] = ("EXPERIMENT, opt-in (model.gemm_engine = 'bf16x3'): nn.Linear GEMMs with M >= 512, the union 1x1 conv and "
                         "the conv3x3 on v_mfma_f32_32x32x16_bf16, each fp32 operand split into three bf16 planes, six cross products, fp32 "
                         "accumulate; error vs fp64 no larger than the exact fp32-MFMA engine's (tests/test_kernels_gpu.py)")
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
            result["workloads"]["dsgdetr_16x12"] = w
            del dm, dsd
            torch.cuda.empty_cache()
        except Exception as e:                           # an extra block must never cost the line
            result["workloads"]["dsgdetr_16x12"] = {"error": repr(e)}
        try:                                             # configs[2] stand-in: the loop of tools/test_STTran.py:75-92 on AG-shaped clips
            result["workloads"]["ag_split_shaped"] = ag_split_sample(device, 256)
        except Exception as e:
            result["workloads"]["ag_split_shaped"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "sttran":
        result["cpu_baseline"] = cpu_baseline(T, N, sd)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        env.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
