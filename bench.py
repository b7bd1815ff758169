#!/usr/bin/env python3
"""bench.py -- frames/sec of the STTran PredCls hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (`STTran.forward`) over one batch of synthetic clips resident in HBM.  Workload of the
line's `value`, AT EVERY N: BASELINE.json configs[1] -- clips of 16 frames x 12 boxes x 2048-d region features (P = 176 pairs
per clip), `--clips-per-step` clips per pass and per GPU (default 64).  Weak scaling: per-GPU work is fixed, so
value(N) / (N x value(1)) IS the scaling efficiency and the driver's SCALE record compares like with like.  (Rounds 1-5
switched the N > 1 `value` to the 64x36 clip; a curve built from those values divided a 64x36 number by a 16x12 one --
VERDICT r5 weak 3.)  `--workload 64x36` selects configs[3]'s clip shape for `value` instead (default 4 clips per pass).

Every run ALSO measures the other clip shape in the same process -- `workloads["64x36"]`: north_star quotes the ">= 6x at 8
GPUs" target on that clip -- and with N > 1 its own N = 1 reference points: rank 0 ALONE on the same per-GPU workload while
the others idle (`one_rank_alone`, for both clip shapes) and the fixed 64-clip strong-scaling set on rank 0 alone.  The
ratios a reader of the driver's record needs are FLAT scalars inside `config` (benchlib/legs_scaling.py::scaling_scalars):
`weak_scaling_efficiency`, `speedup_vs_one_rank`, `scale_64x36_speedup_vs_one_rank`, `strong_64x36_speedup`, ...
The N = 1 default run additionally carries the DSG-DETR model of configs[4] (`workloads["dsgdetr_16x12"]`), a 256-clip sample
of the configs[2] stand-in (`workloads["ag_split_shaped"]`), the one-clip-per-pass rates of the reference's own loop
(`one_clip_per_pass`, `one_clip_coalesced`) and a one-rank RCCL self-test: every BASELINE config has a number in the line.

With N > 1 every rank runs its own clips (whole-clip sharding) and each step's predictions are all-gathered over RCCL inside
the timed region by `lib/distributed.py::PredictionGatherer` (the code the gloo tests cover): issued asynchronously into a
ring of two buffer sets, i.e. under the next forward.  `python bench.py --gpus N` without a launcher (WORLD_SIZE unset)
starts its N ranks ITSELF as fresh child processes -- before this process makes any GPU call -- relays rank 0's JSON line and
exits non-zero if a rank does.

Output (rank 0): ONE compact JSON line on stdout -- every contract field, the roofline of the dominant kernel class (the
fp32 MFMA GEMM: algorithmic 2*M*N*K FLOPs / HIP-event time, measured in a second, instrumented run of the same K steps) with
its dominant kernel's row, the CPU baseline (numpy and torch restatements on this host's cores, bounded sample) and the
scalars of every extra leg; scalars only, < 6 KB, at most 22 keys under `config` / `roofline` / `cpu_baseline`
(benchlib/line.py; the driver keeps the last 8 KB of stdout and 24 keys per dict).  The FULL object (per-kernel-template and
per-shape tables, per-rank records, notes) goes to `--detail` / $BENCH_DETAIL (default bench_detail.json) and to stderr as
one `BENCH_DETAIL {...}` line.

Every timed loop alternates TWO batches with different allocations and different per-frame pair counts (same work), so the
library's index-map and chunk-table caches miss on every step, as in a real loop (`config.layout_cache`); `same_batch` is
the cached loop of rounds 1-3 for comparison.  `pcie_inclusive_overlapped` is the rate when every step's inputs start in
pinned host memory (never `value`).

`--profile-only-batch` runs warm-up + the timed steps of the selected workload and nothing else: the form to put under
`rocprofv3 --kernel-trace --stats`, whose per-kernel averages are then per-step averages.

This file is the CLI, the order of the legs and the line; the legs live in benchlib/ (legs_batch: the timed steps + the
roofline re-run; legs_one_clip; legs_pcie; legs_scaling: gather cost, rank 0 alone, strong scaling, RCCL self-test;
legs_bf16x3: the secondary engine; legs_cpu: the CPU baseline; launch: self-launch of N ranks; line: the compact line).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.launch import launch_ranks, visible_gpu_count  # noqa: E402,F401  (no GPU runtime in there)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=["16x12", "64x36"],
                    help="clip shape of the line's value; default 16x12 (BASELINE configs[1]) at every N")
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
                         "stdout carries one compact line (< 6 KB) only")
    ap.add_argument("--gemm-engine", default="fp32", choices=["fp32", "bf16x3"],
                    help="bf16x3 = the nn.Linear GEMMs with fp32 emulated on the bf16 matrix pipe (three bf16 planes per "
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
    args = ap.parse_args(argv)
    if args.profile_only_batch:
        args.no_cpu_baseline = args.no_roofline = args.no_extra_workloads = True
        args.repeats = 1
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, as fresh children, before anything here touches a GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    import torch
    from benchlib import legs_scaling
    from benchlib.common import SHAPES, Env, flush_c_stdio, make_model
    from benchlib.legs_batch import run_workload
    from benchlib.legs_bf16x3 import bf16x3_block
    from benchlib.legs_cpu import cpu_baseline
    from benchlib.line import compact_line
    env = Env(args)
    rank, world, device = env.rank, env.world, env.device
    # `value` is measured on BASELINE configs[1]'s clip (16x12) at EVERY N, so the driver's per-N values are like-for-like;
    # configs[3]'s clip (64x36, the one north_star quotes the scaling target on) rides in `workloads` with its own rank-0-alone
    # reference and flat ratios in `config`.  `--workload` overrides.
    if args.workload is None:
        args.workload = "16x12"
    other = "16x12" if args.workload == "64x36" else "64x36"
    T, N, cps_default = SHAPES[args.workload]
    cps = args.clips_per_step or cps_default
    model, sd = make_model(args.model, device, args.gemm_engine)

    if args.rccl_selftest:
        if world != 1:
            raise SystemExit("--rccl-selftest is the N = 1 leg (N > 1 runs the same code over the real group)")
        try:
            st = legs_scaling.rccl_selftest(env, model)
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
    default_n1 = extras and world == 1 and args.model == "sttran" and args.workload == "16x12"
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
        result["scaling_note"] = ("weak scaling: every rank runs the same per-GPU workload (the clips of the --gpus 1 line's "
                                  "`value`) on its own clips, so value ~ N x the 1-GPU value unless the host glue or the per-step "
                                  "all-gather contends; config.weak_scaling_efficiency = value / (N x one_rank_alone), measured in "
                                  "this run; config.scale_64x36_* = the same on BASELINE configs[3]'s clip; `strong_scaling` holds "
                                  "the fixed-work legs with their own rank-0-alone basis")
    if "batch_sweep" in main_res:                        # 1 clip / the round 1-2 default / this run's batch, one list
        sweep = []
        if "one_clip_coalesced" in main_res:
            sweep.append({"clips_per_step": 1, "value": main_res["one_clip_coalesced"]["value"],
                          "ms_per_step": main_res["one_clip_coalesced"]["ms_per_step"]})
        sweep += main_res["batch_sweep"]
        sweep.append({"clips_per_step": cps, "value": main_res["value"], "ms_per_step": main_res["ms_per_step"]})
        result["batch_sweep"] = sweep
    for k in ("allgather_ms", "allgather_bytes_per_rank", "one_rank_alone", "one_clip_per_pass", "one_clip_coalesced", "same_batch",
              "two_steps_in_flight", "pcie_inclusive", "pcie_inclusive_overlapped", "roofline", "reference_arithmetic"):
        if k in main_res:
            result[k] = main_res[k]
    # ---- the other BASELINE clip shape in the same run (N > 1: with its own rank-0-alone reference) ----
    if extras and args.model == "sttran":
        steps2 = max(5, min(args.steps, 20))
        w = run_workload(env, model, args.model, other, SHAPES[other][2], steps2, min(args.warmup, 3),
                         roofline=not args.no_roofline, one_clip=(world == 1 and other == "64x36"), alone=True)
        w.pop("unit", None)
        if "roofline" in w:                              # keep the line readable: per-kernel rows, not per-shape
            w["roofline"].pop("by_shape", None)
        result["workloads"] = {other: w}
    # ---- STRONG scaling (fixed work, any N): 64 clips of 64x36, and the Action-Genome-test-split-shaped set (configs[2]'s
    #      stand-in: frames per clip of ag_test_id.pkl, 1..6 pairs per frame) -- model + gather + device evaluator ----
    if extras and args.model == "sttran" and args.gemm_engine == "fp32" and not args.no_strong:
        result["strong_scaling"] = legs_scaling.strong_block(env, model, args)
        if world == 1 and "error" not in result["strong_scaling"]["ag_split_shaped"]:
            result.setdefault("workloads", {})["ag_split_shaped"] = result["strong_scaling"]["ag_split_shaped"]
    # ---- secondary lines: the bf16x3 GEMM engine on both clip shapes (never `value`) ----
    if default_n1 and args.gemm_engine == "fp32":
        for wl in ("16x12", "64x36"):
            try:
                result["workloads"][wl + "_bf16x3"] = bf16x3_block(env, model, args, wl, SHAPES[wl][2], *SHAPES[wl][:2])
            except Exception as e:
                result["workloads"][wl + "_bf16x3"] = {"error": repr(e)}
    # ---- the remaining BASELINE configs, driver-witnessed in the same line (single GPU, default run only) ----
    if default_n1:
        del model
        torch.cuda.empty_cache()
        try:                                             # configs[4]: DSG-DETR (lib/dsg_detr.py, sgdet branch) on the same kernels
            dm, dsd = make_model("dsgdetr", device)
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
    if default_n1 and not args.no_rccl_selftest and not args.no_strong and args.gemm_engine == "fp32":
        # RCCL has no other way to run on a 1-GPU box: a fresh child (never an exec) with a time limit; its verdict rides
        # in the line.  This process's model is gone and its cached blocks were released above.  The child's environment
        # carries no BENCH_DIST_BACKEND / BENCH_FORCE_DEVICE: an "RCCL self-test" over gloo would be no such thing.
        try:
            cenv = {k: v for k, v in os.environ.items() if k not in ("BENCH_DIST_BACKEND", "BENCH_FORCE_DEVICE")}
            cenv.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-selftest", "--gpus", "1"],
                                capture_output=True, text=True, timeout=240, env=cenv)
            last = [l for l in cp.stdout.strip().splitlines() if l.startswith("{")]
            result["rccl_selftest"] = json.loads(last[-1])["rccl_selftest"] if last else {"ok": False, "error": cp.stderr[-300:]}
        except subprocess.TimeoutExpired:
            result["rccl_selftest"] = {"ok": False, "error": "no verdict within 240 s"}
        except Exception as e:
            result["rccl_selftest"] = {"ok": False, "error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "sttran":
        result["cpu_baseline"] = cpu_baseline(T, N, sd)
        if extras and "workloads" in result and other in result["workloads"] and "error" not in result["workloads"][other]:
            # the other clip shape's CPU numbers, on the thread counts that won above (one warm-up + timed forwards)
            c = result["cpu_baseline"]
            result["workloads"][other]["cpu_baseline"] = cpu_baseline(
                *SHAPES[other][:2], sd, budget_s=16.0, threads={"numpy": c["numpy_cores"], "torch": c.get("torch_cores")})
    result["scaling_scalars"] = legs_scaling.scaling_scalars(result, world)
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


def __getattr__(name):
    """`bench.compact_line`, `bench.device_clip`, ... for the tests and tools/experiments that import this file: resolved
    lazily so that importing bench.py (or running its self-launch parent) never loads torch or a GPU runtime."""
    import importlib
    for mod in ("benchlib.line", "benchlib.common", "benchlib.legs_batch", "benchlib.legs_scaling", "benchlib.legs_cpu", "benchlib.legs_bf16x3"):
        m = importlib.import_module(mod)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)


if __name__ == "__main__":
    main()
