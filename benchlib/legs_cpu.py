"""`cpu_baseline`: the CPU restatements of the hot path timed on this host's cores, rank 0 at N = 1 only, on a BOUNDED
sample (a few forwards of one clip).  Two samples of the same clip:

  numpy   oracle/sttran_oracle.py on OpenBLAS (the figure of rounds 1-5; stops scaling at ~8 threads);
  torch   oracle/sttran_torch.py -- F.linear / F.conv2d / baddbmm on oneDNN, the operators the reference's own CPU path
          dispatches to -- at 8 / 64 / all physical cores.

`value` is the FASTER of the two (a stated baseline should be the strongest CPU figure available);
both are reported flat.  The oracle is the checker and the baseline here, never the thing measured or shipped."""
import os
import time

import numpy as np

from nl_vsgg_amd.lib import synthetic as syn


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def _time_forward(fwd, budget_s):
    """one warm-up (BLAS / oneDNN threads, page faults, primitive caches), then up to 3 timed forwards inside the budget"""
    t0 = time.perf_counter()
    fwd()
    first = time.perf_counter() - t0
    runs = []
    while sum(runs) + first < budget_s and len(runs) < 3:
        t0 = time.perf_counter()
        fwd()
        runs.append(time.perf_counter() - t0)
    return (float(np.median(runs)) if runs else first), max(len(runs), 1)


def _numpy_sample(entry, sd, model_kind, budget_s, threads):
    from oracle import sttran_oracle as orc
    fwd = (lambda: orc.dsg_detr_forward(entry, sd)) if model_kind == "dsgdetr" else (lambda: orc.sttran_forward(entry, sd))
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
            med, nruns = _time_forward(fwd, budget_s / len(tries))
        finally:
            if ctx is not None:
                ctx.restore_original_limits()
        if best is None or med < best[0]:
            best = (med, nthr, nruns)
    return best


def _torch_sample(entry, sd, model_kind, budget_s, threads):
    """the torch restatement at 8 / 64 / all physical cores (or `threads`); returns (best (s, threads, runs), {threads: s})"""
    import torch
    from oracle import sttran_torch as ort
    e, w = ort.prepare(entry, sd)                       # tensors made once, outside the timed forwards
    fwd = (lambda: ort.dsg_detr_forward(e, w, prepared=True)) if model_kind == "dsgdetr" else \
          (lambda: ort.sttran_forward(e, w, prepared=True))
    phys = _physical_cores()
    tries = [min(int(threads), phys)] if threads else sorted({min(8, phys), min(64, phys), phys})
    before = torch.get_num_threads()
    best, per = None, {}
    try:
        for nthr in tries:
            torch.set_num_threads(nthr)
            med, nruns = _time_forward(fwd, budget_s / len(tries))
            per[nthr] = med
            if best is None or med < best[0]:
                best = (med, nthr, nruns)
    finally:
        torch.set_num_threads(before)
    return best, per, phys


def cpu_baseline(T, N, sd, budget_s=24.0, model_kind="sttran", threads=None):
    """One clip of T x N per forward.  `threads` = {"numpy": k, "torch": k} pins the thread counts (the second clip shape
    re-uses the winners of the first); default: numpy 8 / 32 / all logical, torch 8 / 64 / all physical, best of each."""
    entry = syn.uniform_clip(11, T, N, mode="sgdet") if model_kind == "dsgdetr" else syn.uniform_clip(11, T, N)
    threads = threads or {}
    ncpu = os.cpu_count() or 1
    n_med, n_thr, n_runs = _numpy_sample(entry, sd, model_kind, budget_s / 2, threads.get("numpy"))
    out = {"unit": "frames/s", "host_cores": ncpu, "kind": "port",
           "numpy_value": T / n_med, "numpy_cores": n_thr}
    what = "DSG-DETR sgdet" if model_kind == "dsgdetr" else "STTran PredCls"
    try:
        (t_med, t_thr, t_runs), per, phys = _torch_sample(entry, sd, model_kind, budget_s / 2, threads.get("torch"))
        out.update(torch_value=T / t_med, torch_cores=t_thr, host_physical_cores=phys)
        for k, s in per.items():
            tag = "all" if k == phys and k not in (8, 64) else str(k)
            out[f"torch_value_{tag}_threads"] = T / s
    except Exception as e:                              # the second sample must never cost the first
        out["torch_error"] = repr(e)[:200]
        t_med = None
    if t_med is not None and t_med < n_med:
        out.update(value=T / t_med, cores=t_thr, impl="torch")
        med, nruns, thr = t_med, t_runs, t_thr
    else:
        out.update(value=T / n_med, cores=n_thr, impl="numpy")
        med, nruns, thr = n_med, n_runs, n_thr
    out["sample"] = (f"{nruns} fwd of one {T}x{N} clip ({what}), fp32, {out['impl']} restatement at {thr} threads, "
                     f"median {med:.3f} s/clip")
    return out
