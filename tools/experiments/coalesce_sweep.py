#!/usr/bin/env python3
"""One clip per call, the reference's loop body, under `model.coalesce = K` on L lanes: frames/s over K x L (16x12 and 64x36
clips, a different entry on every call as in bench.py's one-clip leg)."""
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran  # noqa: E402

dev = torch.device("cuda:0")
m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=bench.CLASSES,
           enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}, strict=False)
m.check_indices = False
for T, N, Ks, n in ((16, 12, (1, 4, 8, 16, 32, 64), 768), (64, 36, (1, 2, 4, 8), 96)):
    gen = torch.Generator(device=dev).manual_seed(5)
    ones = [bench.device_clip(T, N, gen, dev, shifted=(i == 1)) for i in range(2)]
    P, B = int(ones[0]["pair_idx"].shape[0]), int(ones[0]["features"].shape[0])
    for lanes in (1, 2, 3):
        row = []
        for K in Ks:
            m.lanes, m.coalesce = lanes, K
            m.reserve(K * P + 8, K * B + 8)
            depth = m.pipeline_depth

            def loop(cnt):
                pending = collections.deque()
                for i in range(cnt):
                    pending.append(m.forward_async(dict(ones[i % 2])))
                    if len(pending) == depth:
                        m.join(pending.popleft())
                while pending:
                    m.join(pending.popleft())
            loop(max(2 * depth, 16))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(n)
            torch.cuda.synchronize()
            row.append(f"K={K}: {T * n / (time.perf_counter() - t0):8.0f}")
        m.sync_check()
        print(f"{T}x{N} lanes={lanes}  " + "  ".join(row), flush=True)
