"""The leg that makes the line's `value`: warm-up, then EXACTLY `steps` timed steps of the hot path between barrier +
synchronize (max over ranks), and -- behind it, never inside it -- the instrumented re-run that prices the dominant
kernel class against the fp32-MFMA roofline."""
import json
import os
import time

import numpy as np
import torch

from nl_vsgg_amd.lib.distributed import PredictionGatherer, pack_predictions
from nl_vsgg_amd.lib.sttran import pack_clips

from .common import FP32_MFMA_PEAK_TFLOPS, ROOT, SHAPES, make_batch


class Workload:
    """One (model, clip shape, clips per step) workload of one rank: the two alternating batches, the step, the gatherer."""

    def __init__(self, env, model, model_kind, workload, cps, steps, rotate=True):
        self.env, self.model, self.model_kind, self.name, self.cps, self.steps = env, model, model_kind, workload, cps, steps
        self.T, self.N, _ = SHAPES[workload]
        seed = 1234 if workload == "16x12" else 4321
        # TWO batches alternate through every loop: different allocations (every tensor of every clip) and different
        # per-frame pair counts (clip 0 of batch 1 is "shifted": one object moved between two interior frames -- same frames,
        # boxes, pairs and window tokens, i.e. the same work).  A real loop hands over new tensors with new frame counts on
        # every call (tools/test_STTran.py:81-84), so the library's index-map cache and chunk-table cache MISS on every step:
        # the host-side build_layout and the two staged uploads are inside the timed region.  `same_batch_leg` re-forwards
        # ONE batch (both caches hit) to show what that costs.
        self.batches = [make_batch(env, model_kind, self.T, self.N, cps, seed + 97 * j, shifted=(j == 1 and rotate))
                        for j in range(2 if rotate else 1)]
        self.clips = self.batches[0]
        self.P = sum(int(c["pair_idx"].shape[0]) for c in self.clips)
        self.B = sum(int(c["features"].shape[0]) for c in self.clips)
        model.reserve(self.P, self.B)
        self._turn = 0
        # per-clip predictions of every rank: one fixed-size RCCL all-gather per step (PredictionGatherer)
        self.gatherer = PredictionGatherer(self.P, cps, cols=26, device=env.device, depth=2) if env.world > 1 else None
        self.clip_ids = [env.rank * cps + i for i in range(cps)]
        self.clip_pairs = [int(c["pair_idx"].shape[0]) for c in self.clips]          # identical for both batches
        self.frames_per_step = env.world * cps * self.T

    def forward_batch(self, which=None):
        # The batch is formed HERE, inside the step, from the separate per-clip dicts a producer hands over one at a time
        # (tools/test_STTran.py:81-84): pack_clips(copy=False) passes the clips' own tensors to the library as per-clip
        # pointer tables -- nothing is concatenated, so no copy hides outside the timed region.
        if which is None:
            which = self._turn % len(self.batches)
            self._turn += 1
        b = self.batches[which]
        return self.model(pack_clips(b, copy=False)) if self.cps > 1 else self.model(dict(b[0]))

    def step(self):
        pred = self.forward_batch()
        if self.gatherer is not None:
            self.gatherer.submit(pack_predictions(pred, out=self.gatherer.payload()), self.clip_ids, self.clip_pairs)
        return pred

    def release(self):
        self.batches = self.clips = None
        torch.cuda.empty_cache()


def timed_steps(w, warmup, repeats, graph=False):
    """W untimed warm-up steps, then `repeats` back-to-back regions of EXACTLY `w.steps` steps, each bracketed by barrier +
    torch.cuda.synchronize() (`Env.barrier`) and reduced with MAX over the ranks; the value is the MEDIAN region (a 20-step
    region is 0.6 s: one region alone moves +-1 % with the box's clocks)."""
    env, model = w.env, w.model
    for _ in range(warmup):
        w.step()
    env.barrier(w.gatherer)
    run = w.step
    if graph and env.world == 1:
        side = torch.cuda.Stream(env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            w.step(); w.step()                              # warm every lazy path on the capture stream
        torch.cuda.current_stream(env.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gpred = w.step()

        def run():
            g.replay()
            return gpred
        run(); torch.cuda.synchronize()
    times = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(w.steps):
            pred = run()
        env.barrier(w.gatherer)
        times.append(env.max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(times))
    assert torch.isfinite(pred["attention_distribution"]).all()
    model.sync_check()                  # device-side flags (clamped indices, skipped DSG-DETR sequences) raise here
    T, N, cps, world = w.T, w.N, w.cps, env.world
    return pred, {
        "value": w.frames_per_step * w.steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / w.steps,
        "steps": w.steps, "warmup": warmup, "repeats": [w.frames_per_step * w.steps / t for t in times],
        "timed_seconds": sum(times),
        "config": {"workload": (f"synthetic {T} frames x {N} boxes x 2048-d region features, STTran PredCls forward "
                                f"(enc 1 / dec 3 layers, d=1936), inputs resident in HBM") if w.model_kind == "sttran" else
                               (f"synthetic {T} frames x {N} boxes x 2048-d region features, DSG-DETR sgdet forward "
                                f"(1 spatial + 3 temporal encoder layers, d=1936), inputs resident in HBM"),
                   "clips_per_step": cps, "hip_graph": bool(graph and world == 1), "frames_per_clip": T,
                   "batch": (f"formed inside every timed step from {cps} separate per-clip entries, by pointer "
                             f"(pack_clips(copy=False): per-clip pointer tables, no concatenation)") if cps > 1 else "one clip",
                   "boxes_per_frame": N, "pairs_per_step": w.P,
                   "sharding": f"whole clips, {world} rank(s), one RCCL all-gather of [pairs, 26] prediction rows per step "
                               f"(asynchronous, ring of 2 buffer sets)" if world > 1 else "single GPU",
                   "layout_cache": ("miss every step: two batches of different allocations and different per-frame pair counts "
                                    "alternate, so build_layout and both staged uploads run inside every timed step")
                                   if len(w.batches) > 1 else "hit (one batch re-forwarded)"},
    }


def same_batch_leg(w, ms_per_step):
    """the loop of rounds 1-3 for comparison: ONE batch re-forwarded, so the index-map and chunk-table caches hit"""
    for _ in range(2):
        w.forward_batch(0)
    torch.cuda.synchronize()
    n0 = max(4, min(w.steps, 20))
    t0 = time.perf_counter()
    for _ in range(n0):
        w.forward_batch(0)
    torch.cuda.synchronize()
    dt0 = (time.perf_counter() - t0) / n0
    return {"value": w.frames_per_step / dt0, "ms_per_step": 1e3 * dt0, "steps": n0,
            "layout_cache": "hit", "delta_ms_per_step_vs_value": ms_per_step - 1e3 * dt0}


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


def roofline_leg(w):
    """Roofline of the dominant kernel class: an instrumented re-run of the same K steps.  The library brackets every launch
    site with HIP events on the stream the kernels are launched on (`sttran_profile_enable`); FLOPs are algorithmic 2*M*N*K
    (unpadded) per launch site."""
    env, model = w.env, w.model
    model.profile(True)
    for _ in range(w.steps):
        w.step()
    prof = model.profile_read()
    entries = model.profile_entries()
    model.profile(False, reset=False)
    env.barrier(w.gatherer)
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
    pmc = sorted(p for p in os.listdir(pdir) if p.endswith(f"pmc_traffic_{w.name}.json")) if os.path.isdir(pdir) else []
    if pmc and w.model_kind == "sttran":
        with open(os.path.join(pdir, pmc[-1])) as f:
            pj = json.load(f)
        cls = pj["classes"].get("gemm")
        # per-launch bytes scale with the batch: only a PMC file taken at this run's clips per step applies
        # (files older than round 2's r2_d carry no `clips_per_step`: they were taken at 16 / 1 clips)
        pmc_cps = pj.get("clips_per_step") or {"16x12": 16, "64x36": 1}[w.name]
        if cls and pmc_cps == w.cps:
            traffic, traffic_src, traffic_commit = cls["hbm_bytes_per_launch"], f"profiles/{pmc[-1]}", pj.get("commit")
    by_kernel, by_shape = by_kernel_tables(entries, prof["forwards"])
    dom = next((r for r in by_kernel if r["class"] == "gemm" and "tflops" in r), None)      # sorted by time per step
    return {
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


def reference_arithmetic(T, N, frames_per_s_per_gpu):
    """the whole forward against the MFMA ceiling of the REFERENCE's arithmetic (what lib/sttran.py executes
    per clip, before this implementation's de-duplication / dead-row elimination)"""
    n, Pc = N - 1, T * (N - 1)
    dec_tok = 2 * n * (T - 1)
    flop_clip = (Pc * (102_238_208 + 45_844_480 + 100_672) + 3 * dec_tok * 45_844_480
                 + Pc * 7_744 * n + 3 * dec_tok * 7_744 * 2 * n)
    eq = frames_per_s_per_gpu * (flop_clip / T) / 1e12
    return {"gflop_per_frame": flop_clip / T / 1e9, "tflops_equivalent_per_gpu": eq,
            "frac_of_fp32_mfma_peak": eq / FP32_MFMA_PEAK_TFLOPS}


def run_workload(env, model, model_kind, workload, cps, steps, warmup, *, graph=False, roofline=True, one_clip=False,
                 pcie=False, repeats=1, alone=False, rotate=True, same_batch=True):
    """One workload on this rank: the timed steps, then the optional legs (each in its own module)."""
    from . import legs_one_clip, legs_pcie, legs_scaling
    w = Workload(env, model, model_kind, workload, cps, steps, rotate=rotate)
    pred, res = timed_steps(w, warmup, repeats, graph=graph)
    world = env.world
    if len(w.batches) > 1 and world == 1 and not graph and same_batch:
        res["same_batch"] = same_batch_leg(w, res["ms_per_step"])
    if world > 1:
        res.update(legs_scaling.allgather_cost(w, pred))
        if alone:
            one = legs_scaling.one_rank_alone(w)
            if one is not None:
                res["one_rank_alone"] = one
    if one_clip and cps > 1 and world == 1:
        res.update(legs_one_clip.one_clip_legs(w))
        res["two_steps_in_flight"] = legs_one_clip.two_steps_in_flight(w)
        sweep = legs_one_clip.batch_sweep(w)
        if sweep:
            res["batch_sweep"] = sweep
    if pcie and world == 1:
        res.update(legs_pcie.pcie_legs(w, pcie))
    if roofline:
        res["roofline"] = roofline_leg(w)
    if model_kind == "sttran":
        res["reference_arithmetic"] = reference_arithmetic(w.T, w.N, res["value"] / world)
    w.release()
    return res
