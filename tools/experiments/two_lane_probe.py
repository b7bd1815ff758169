#!/usr/bin/env python3
"""Experiment: does running independent sub-batches on separate HIP streams (one handle each) recover the kernel tails
(stream-K parking + fix-up launches, epilogues, attention / LayerNorm launches) of the single-stream forward?
    python tools/experiments/two_lane_probe.py [--lanes 1,2,3,4] [--clips 16]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran, pack_clips  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--clips", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--workload", default="16x12")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    T, N, _ = bench.SHAPES[a.workload]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}
    gen = torch.Generator(device=dev).manual_seed(1234)
    clips = [bench.device_clip(T, N, gen, dev) for _ in range(a.clips)]
    for lanes in [int(x) for x in a.lanes.split(",")]:
        models, batches, streams = [], [], []
        per = a.clips // lanes
        for l in range(lanes):
            m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=bench.CLASSES,
                       enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
            m.eval(); m.check_indices = False
            m.load_state_dict(sd, strict=False)
            models.append(m)
            cs = clips[l * per:(l + 1) * per]
            batches.append(pack_clips(cs) if len(cs) > 1 else cs[0])
            streams.append(torch.cuda.Stream(dev))

        def step():
            for m, b, s in zip(models, batches, streams):
                with torch.cuda.stream(s):
                    m(dict(b))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"lanes {lanes}: {1e3 * dt:.3f} ms/step  {per * lanes * T / dt:.0f} frames/s", flush=True)
        del models, batches, streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
