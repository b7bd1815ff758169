#!/usr/bin/env python3
"""Half-period timeline of the ping-pong mask-conv kernel (experiment build with -DSTTRAN_MC_TRACE, loaded through STTRAN_LIB):
s_memtime stamps of workgroups 0..3, both groups: cycles of the work and of the barrier wait in every half-period."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nl_vsgg_amd import _native as nat  # noqa: E402
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran, pack_clips  # noqa: E402

m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=bench.CLASSES,
           enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}, strict=False)
m.check_indices = False
gen = torch.Generator(device="cuda").manual_seed(1)
clips = [bench.device_clip(16, 12, gen, torch.device("cuda:0")) for _ in range(64)]
for _ in range(3):
    m(pack_clips(clips, copy=False))
torch.cuda.synchronize()
lib = nat.load()
buf = (C.c_ulonglong * (4 * 2 * 32 * 3))()
assert lib.sttran_debug_mc_trace(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(4, 2, 32, 3)
for wg in range(2):
    print(f"workgroup {wg}: pair | start, conv cycles (350 MFMAs = 22 400), masks -> LDS + epilogue cycles, barrier + loop")
    for it in range(10):
        a, b, c = t[wg, 0, it]
        nxt = t[wg, 0, it + 1, 0]
        print(f"  pair {it:2d} | {a - t[wg, 0, 0, 0]:8d} {b - a:7d} {c - b:7d} {nxt - c:7d}")
fine = (C.c_ulonglong * 64)()
if hasattr(lib, "sttran_debug_mc_fine") and lib.sttran_debug_mc_fine(fine) == 0:
    f = np.array(fine, dtype=np.int64)
    names = ["params read", "activation + writes issued", "writes visible (fence)", "reads + max + stores issued", "fence"]
    print("epilogue of one pair (workgroup 0, group 0), cycles per segment:")
    for q in range(4):
        seg = [int(f[1 + 5 * q + k] - f[5 * q + k]) for k in range(5)]
        print(f"  round {q}: " + ", ".join(f"{n} {v}" for n, v in zip(names, seg)))
    print("  write-out", int(f[21] - f[20]), " total", int(f[21] - f[0]))
