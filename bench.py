#!/usr/bin/env python3
"""bench.py -- frames/sec of the STTran PredCls hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (`STTran.forward`) over one batch of synthetic clips resident
in HBM.  Workload (default, BASELINE.json configs[1]): clips of 16 frames x 12 boxes x 2048-d
region features (P = 176 pairs per clip), `--clips-per-step` clips per pass (default 16).  `--workload 64x36`
selects configs[3]'s clip shape.  With N > 1 every rank runs its own clips (whole-clip sharding,
weak scaling) and the per-step predictions are all-gathered over RCCL inside the timed region.

One JSON line is printed by rank 0; it carries the roofline of the dominant kernel (the fp32 MFMA
GEMM: algorithmic 2*M*N*K FLOPs / HIP-event time, measured in a second, instrumented run of the
same K steps) and a CPU baseline (the numpy oracle on this host's cores, bounded sample).
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
from nl_vsgg_amd.lib.distributed import pack_predictions  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran, pack_clips  # noqa: E402

CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


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
    return {"value": T / med, "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": f"{nruns} forward(s) of one {T}x{N} clip, numpy/BLAS fp32 oracle, best of 8/32/{ncpu} BLAS "
                      f"threads, median {med:.3f} s/clip"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="16x12", choices=["16x12", "64x36"])
    ap.add_argument("--clips-per-step", type=int, default=0, help="0 = default for the workload")
    ap.add_argument("--model", default="sttran", choices=["sttran", "dsgdetr"],
                    help="dsgdetr = BASELINE.json configs[4]: lib/dsg_detr.py (sgdet branch) on the same kernels")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and replay it: the forward only "
                         "enqueues on the caller's stream, so it is capturable once the layout is cached")
    ap.add_argument("--pcie", action="store_true",
                    help="also report the rate when every step first copies its inputs from pinned host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one process per GPU; BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only so the N>1 code path can be
    # smoke-tested on a single-GPU box (all ranks on device 0, gloo instead of RCCL)
    local = int(os.environ.get("BENCH_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm
        dist.init_process_group(backend)

    if args.graph and args.model == "dsgdetr":
        raise SystemExit("--graph: the DSG-DETR forward reads labels / pair_idx back to build its class sequences "
                         "(one small D2H + sync per call), so it cannot be captured")
    T, N = (16, 12) if args.workload == "16x12" else (64, 36)
    cps = args.clips_per_step or (16 if args.workload == "16x12" else 1)
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
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    clips = [device_clip(T, N, gen, device) for _ in range(cps)]
    if args.model == "dsgdetr":                   # sgdet entry: detector boxes, class distribution, scores
        for c in clips:
            B = c["features"].shape[0]
            xy = torch.rand(B, 2, device=device, generator=gen) * 300
            wh = torch.rand(B, 2, device=device, generator=gen) * 150 + 10
            c["boxes"] = torch.cat([torch.arange(T, device=device).repeat_interleave(N)[:, None].float(), xy, xy + wh], 1)
            c["distribution"] = torch.softmax(torch.randn(B, 36, device=device, generator=gen), 1)
            c["scores"] = c["distribution"].max(1).values
            c["im_idx"] = c["im_idx"].long()
    batch = pack_clips(clips) if cps > 1 else clips[0]
    P = int(batch["pair_idx"].shape[0])
    model.reserve(P, int(batch["features"].shape[0]))
    # per-clip predictions of every rank: one fixed-size RCCL all-gather per step, issued asynchronously into
    # one of two buffers so that it runs on RCCL's stream under the next step's forward (it is waited for
    # two steps later and, for the last steps, in barrier(): all of them finish inside the timed region)
    gathered = [torch.empty((world * P, 26), device=device) for _ in range(2)] if world > 1 else None
    inflight = [None, None]
    turn = [0]

    def step():
        # a fresh dict per call: forward() writes its outputs into the entry (sgdet replaces `distribution`)
        pred = model(dict(batch))
        if world > 1:
            k = turn[0]
            if inflight[k] is not None:
                inflight[k][0].wait()
            rows = pack_predictions(pred)
            inflight[k] = (dist.all_gather_into_tensor(gathered[k], rows, async_op=True), rows)
            turn[0] = k ^ 1
        return pred

    def barrier():
        if world > 1:
            for k in range(2):
                if inflight[k] is not None:
                    inflight[k][0].wait()
                    inflight[k] = None
            dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    run = step
    if args.graph and world == 1:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            step(); step()                              # warm every lazy path on the capture stream
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gpred = step()

        def run():
            graph.replay()
            return gpred
        run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred = run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(pred["attention_distribution"]).all()

    frames_per_step = world * cps * T
    result = {
        "metric": "frames/sec (PredCls inference)" if args.model == "sttran" else "frames/sec (SGDet inference, DSG-DETR)",
        "value": frames_per_step * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"synthetic {T} frames x {N} boxes x 2048-d region features, STTran PredCls forward "
                                f"(enc 1 / dec 3 layers, d=1936), inputs resident in HBM") if args.model == "sttran" else
                               (f"synthetic {T} frames x {N} boxes x 2048-d region features, DSG-DETR sgdet forward "
                                f"(1 spatial + 3 temporal encoder layers, d=1936), inputs resident in HBM"),
                   "clips_per_step": cps, "hip_graph": bool(args.graph and world == 1), "frames_per_clip": T, "boxes_per_frame": N, "pairs_per_step": P,
                   "sharding": f"whole clips, {world} rank(s), RCCL all-gather of predictions" if world > 1
                               else "single GPU"},
    }

    # ---- the same clip shape, ONE clip per pass (the reference's own batch size; not `value`) -------------
    if cps > 1 and world == 1:
        one = clips[0]
        for _ in range(3):
            model(dict(one))
        torch.cuda.synchronize()
        n1 = max(2 * args.steps, 20)
        t0 = time.perf_counter()
        for _ in range(n1):
            model(dict(one))
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / n1
        result["one_clip_per_pass"] = {"value": T / dt1, "unit": "frames/s", "ms_per_step": 1e3 * dt1,
                                       "note": "same clip shape with clips_per_step = 1 (latency-bound: one clip cannot "
                                               "fill 256 CUs)"}
        model(dict(batch))                              # restore the cached layout of the batch

    # ---- PCIe-inclusive rate (never `value`): inputs start in pinned host memory each step ----------
    if args.pcie and world == 1:
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
        for _ in range(max(args.steps // 2, 3)):
            step_h2d()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(args.steps // 2, 3)
        result["pcie_inclusive"] = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
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
        n_over = max(args.steps, 6)
        t0 = time.perf_counter()
        pipelined(n_over)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_over
        result["pcie_inclusive_overlapped"] = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                               "note": "H2D of step i+1 on a copy stream under the forward of step i"}

    # ---- roofline of the dominant kernel: instrumented re-run of the same K steps -------------
    if not args.no_roofline:
        model.profile(True)
        for _ in range(args.steps):
            step()
        prof = model.profile_read()
        model.profile(False, reset=False)
        g = prof["gemm"]
        tot_ms = sum(v["ms"] for k, v in prof.items() if isinstance(v, dict))
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        # HBM-side traffic of the same kernel class: PMC counters cannot be read from inside this process;
        # tools/pmc_traffic.py turns the two rocprofv3 --pmc passes of THIS command (FETCH_SIZE x2 per the
        # gfx950 correction, WRITE_SIZE) into profiles/*_pmc_traffic_<workload>.json, picked up here.
        traffic, traffic_src = None, None
        pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles"))
                     if p.endswith(f"pmc_traffic_{args.workload}.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if pmc and args.model == "sttran":
            with open(os.path.join(ROOT, "profiles", pmc[-1])) as f:
                cls = json.load(f)["classes"].get("gemm")
            if cls:
                traffic, traffic_src = cls["hbm_bytes_per_launch"], f"profiles/{pmc[-1]}"
        result["roofline"] = {
            "kernel": "gemm_sk_kernel + gemm_fixup_kernel (fp32 MFMA 32x32x2, all nn.Linear / conv3x3 launches)",
            "bound": "mfma", "achieved": ach,
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes per launch (L2 fabric reads x2 + writes)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": g["bytes"] / max(g["launches"], 1),
            "launches_per_step": g["launches"] / max(prof["forwards"], 1),
            "avg_launch_us": 1e3 * g["ms"] / max(g["launches"], 1),
            "share_of_device_time": g["ms"] / tot_ms if tot_ms else None,
            "per_class_ms_per_step": {k: v["ms"] / max(prof["forwards"], 1) for k, v in prof.items()
                                      if isinstance(v, dict) and v["launches"]},
            "per_class_tflops": {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in prof.items()
                                 if isinstance(v, dict) and v["ms"] > 0 and v["flops"] > 0},
        }
    if args.model == "sttran":
        # the whole forward against the MFMA ceiling of the REFERENCE's arithmetic (what lib/sttran.py
        # executes per clip, before this implementation's de-duplication / dead-row elimination)
        n, Pc = N - 1, T * (N - 1)
        dec_tok = 2 * n * (T - 1)
        flop_clip = (Pc * (102_238_208 + 45_844_480 + 100_672) + 3 * dec_tok * 45_844_480
                     + Pc * 7_744 * n + 3 * dec_tok * 7_744 * 2 * n)
        eq = result["value"] / world * (flop_clip / T) / 1e12
        result["reference_arithmetic"] = {"gflop_per_frame": flop_clip / T / 1e9, "tflops_equivalent_per_gpu": eq,
                                          "frac_of_fp32_mfma_peak": eq / FP32_MFMA_PEAK_TFLOPS}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "sttran":
        result["cpu_baseline"] = cpu_baseline(T, N, sd)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
