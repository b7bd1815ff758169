#!/usr/bin/env python3
"""what does the co-resident workgroup do while its partner is between two tiles?
One launch of the 128 x 176 GEMM tile with per-tile s_memtime stamps (experiment build of the library,
make EXTRA=-DSTTRAN_GEMM_EXPERIMENT, loaded through STTRAN_LIB; STTRAN_T16_ABLATE=9):
    STTRAN_LIB=<exp build> python tools/experiments/t16_trace.py 21120,1936,1936
Per CU (XCC, SE, CU from HW_ID) with two resident workgroups: the GAP of a workgroup = last MFMA of tile i -> first MFMA
of tile i + 1 (epilogue + next prologue); how much of that gap its partner spent inside ITS main loop."""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["STTRAN_T16_ABLATE"] = "9"
from nl_vsgg_amd import _native  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1].split(","))
lib = _native.load()
raw = C.CDLL(_native.LIB_PATH)
p = lambda t: C.c_void_p(t.data_ptr())
Kp = (K + 31) // 32 * 32
A = torch.randn(M + 1, Kp, device="cuda"); W = torch.zeros(N, Kp, device="cuda"); W[:, :K] = torch.randn(N, K, device="cuda")
b = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda"); R = torch.randn(M, N, device="cuda")
run = lambda: lib.sttran_debug_gemm_padded(p(A), Kp, None, p(W), Kp, p(b), p(R), p(Cc), M, N, K, 0, 5, None)
for _ in range(20):
    run()
torch.cuda.synchronize()
cap = 1 << 14
buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
raw.sttran_debug_t16_trace.argtypes = [C.c_void_p, C.c_uint]
assert raw.sttran_debug_t16_trace(C.c_void_p(buf.data_ptr()), cap) == 0
run()
torch.cuda.synchronize()
n = C.c_uint(0)
raw.sttran_debug_t16_trace_count(C.byref(n))
raw.sttran_debug_t16_trace(None, 0)
rec = buf.cpu().numpy().reshape(-1, 8)[:min(n.value, cap)].astype(np.uint64)
hw, xcc = rec[:, 0], rec[:, 1] & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
nsteps = rec[:, 3] >> 32
T = rec[:, 4:8].astype(np.int64)
# __builtin_readcyclecounter() counts shader clocks of the wave's XCD (every XCD has its own counter and DVFS state): stamps
# are only compared inside ONE CU, each CU's first tile start is its time 0, and a tick is priced at the clock the PMC
# passes report for this kernel (GRBM_GUI_ACTIVE / duration: 2.3 GHz) -- good to a few per cent
GHZ = 2.3
T = T.astype(np.float64)
key_all = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
for k in np.unique(key_all):
    m = key_all == k
    T[m] -= T[m].min()
T = T / (GHZ * 1e3)                                      # ticks -> microseconds
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
launch_us = e0.elapsed_time(e1) * 100.0                  # the ABL = 9 build (it waits for its stores): us per launch
by_cu = defaultdict(lambda: defaultdict(list))
for i in range(len(rec)):
    by_cu[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))][int(rec[i, 2])].append((T[i], int(nsteps[i])))
print(f"{M}x{N}x{K}: {len(rec)} tile segments, {len(by_cu)} CUs seen, workgroups per CU: "
      f"{sorted(set(len(v) for v in by_cu.values()))}")
gaps, cover, offs, pro, epi = [], [], [], [], []
for key, wgs in by_cu.items():
    if len(wgs) != 2:
        continue
    (a, sa), (b_, sb) = [(k, sorted(v, key=lambda x: x[0][0])) for k, v in wgs.items()]
    for me, other in ((sa, sb), (sb, sa)):
        loops = [(t[1], t[2]) for t, _ in other]         # partner's main-loop intervals [first MFMA, last MFMA]
        for (t_i, _), (t_n, _) in zip(me[:-1], me[1:]):
            g0, g1 = t_i[2], t_n[1]                      # my gap: last MFMA of this tile -> first MFMA of the next
            if g1 <= g0:
                continue
            c = sum(max(0.0, min(g1, l1) - max(g0, l0)) for l0, l1 in loops)
            gaps.append(g1 - g0); cover.append(c / (g1 - g0))
            epi.append(t_i[3] - t_i[2]); pro.append(t_n[1] - t_n[0])
    offs.append(abs(sa[0][0][0] - sb[0][0][0]))
gaps, cover = np.array(gaps), np.array(cover)
print(f"  gaps between two tiles of a workgroup: {len(gaps)}, mean {gaps.mean():.1f} us (epilogue to last store ack {np.mean(epi):.1f} us, "
      f"next prologue {np.mean(pro):.1f} us; the two overlap: stores are not waited for)")
print(f"  fraction of a gap during which the co-resident partner is inside its main loop: mean {cover.mean():.2f}, "
      f"median {np.median(cover):.2f}, <10 % covered: {(cover < 0.1).mean():.2f} of the gaps, >90 % covered: {(cover > 0.9).mean():.2f}")
print(f"  start offset between the two workgroups of a CU: mean {np.mean(offs):.1f} us, median {np.median(offs):.1f} us")
# where the launch's time goes, per CU: first start .. last end of its two workgroups
ends = np.array([max(t[3] for wg in v.values() for t, _ in wg) for v in by_cu.values()])
first_done = np.array([min(max(t[3] for t, _ in wg) for wg in v.values()) for v in by_cu.values() if len(v) == 2])
last_done = np.array([max(max(t[3] for t, _ in wg) for wg in v.values()) for v in by_cu.values() if len(v) == 2])
dur = {blk: sum(t[2] - t[1] for t, ns in wg if ns == 61) / max(1, sum(1 for t, ns in wg if ns == 61)) for v in by_cu.values() for blk, wg in v.items()}
d = np.array([x for x in dur.values() if x > 0])
print(f"  launch {launch_us:.0f} us (instrumented build); a CU's last workgroup ends at {ends.mean():.0f} us on average (min {ends.min():.0f}, max {ends.max():.0f}): "
      f"{(ends.max() - ends.mean()) / ends.max():.3f} of the launch is CUs waiting for the slowest one")
print(f"  the FIRST workgroup of a CU to finish does so {np.mean(last_done - first_done):.0f} us before its partner (max {np.max(last_done - first_done):.0f}): "
      f"the partner then runs alone")
print(f"  main loop of a whole 61-step tile: mean {d.mean():.0f} us, min {d.min():.0f}, max {d.max():.0f}, p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f} "
      f"(two workgroups share the CU's matrix pipes: an even split would give every tile the same time)")
# the two workgroups of a CU: the one dispatched first (lower blk inside its XCD's block of 64) against the other
ratios, fav_first = [], 0
for v in by_cu.values():
    if len(v) != 2:
        continue
    (b0, w0), (b1, w1) = sorted(v.items())
    m0 = [t[2] - t[1] for t, ns in w0 if ns == 61]
    m1 = [t[2] - t[1] for t, ns in w1 if ns == 61]
    if m0 and m1:
        ratios.append(np.mean(m1) / np.mean(m0))
        fav_first += (b1 - b0) == 32
ratios = np.array(ratios)
print(f"  per CU: whole-tile main-loop time of the LATER-dispatched workgroup / of the earlier one: mean {ratios.mean():.3f}, "
      f"p10 {np.percentile(ratios, 10):.3f}, p50 {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, min {ratios.min():.3f}, max {ratios.max():.3f} "
      f"({fav_first} of {len(ratios)} pairs are blk, blk + 32)")
k = next(k for k, v in by_cu.items() if len(v) == 2)
print(f"  one CU {k} (us: start, first MFMA, last MFMA, stores acked | K-steps):")
for blk, v in by_cu[k].items():
    for t, ns in sorted(v, key=lambda x: x[0][0])[:5]:
        print(f"    wg {blk:4d}: {t[0]:8.1f} {t[1]:8.1f} {t[2]:8.1f} {t[3]:8.1f} | {ns}")
