#!/usr/bin/env python3
"""Experiment: the reference's loop forwards ONE clip per call; can K clips in flight on K HIP streams
hide a launch's fixed cost (ramp, prologue, epilogue: ~9 of ~17 us per small-M GEMM) under another clip's MFMAs?
Probe without any library change: K handles (own workspace each), K streams, clip i -> lane i % K.
    python tools/experiments/lanes_probe.py [--lanes 1,2,3,4] [--workload 16x12] [--calls 200]
Also prints the host's enqueue time per forward (the loop without the final synchronisation)."""
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
from nl_vsgg_amd.lib.sttran import STTran  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1,2,3,4")
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--workload", default="16x12")
    ap.add_argument("--pack", type=int, default=1, help="--api: clips per call (pack_clips(copy=False)); 1 = single clips")
    ap.add_argument("--api", action="store_true", help="ONE handle with `model.lanes = K` (forward_async / join) instead of K handles")
    ap.add_argument("--join", default="lag", choices=["lag", "end"], help="--api: join call i - K + 1 after submitting call i, or only at the end")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    T, N, _ = bench.SHAPES[a.workload]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()}
    gen = torch.Generator(device=dev).manual_seed(1234)
    clips = [bench.device_clip(T, N, gen, dev, shifted=bool(i & 1)) for i in range(8)]
    if a.api:
        import collections
        m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=bench.CLASSES,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
        m.eval(); m.check_indices = False
        m.load_state_dict(sd, strict=False)
        packs = [[bench.device_clip(T, N, gen, dev, shifted=(j == 0 and b == 1)) for j in range(a.pack)] for b in range(2)] if a.pack > 1 else None
        for lanes in [int(x) for x in a.lanes.split(",")]:
            m.lanes = lanes

            def run(n):
                pending = collections.deque()
                for i in range(n):
                    if a.pack > 1:
                        from nl_vsgg_amd.lib.sttran import pack_clips
                        entry = pack_clips(packs[i % len(packs)], copy=False)
                    else:
                        entry = dict(clips[i % len(clips)])
                    pending.append(m.forward_async(entry))
                    if a.join == "lag" and len(pending) == lanes:
                        m.join(pending.popleft())
                m.join()
            run(4 * lanes)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(a.calls)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"api lanes {lanes} join={a.join} pack={a.pack}: {1e3 * (t2 - t0) / a.calls:.3f} ms/call  {a.calls * a.pack * T / (t2 - t0):.0f} frames/s   "
                  f"(host enqueue {1e3 * (t1 - t0) / a.calls:.3f} ms/clip)", flush=True)
        return
    for lanes in [int(x) for x in a.lanes.split(",")]:
        models, streams = [], []
        for l in range(lanes):
            m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=bench.CLASSES,
                       enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
            m.eval(); m.check_indices = False
            m.load_state_dict(sd, strict=False)
            models.append(m)
            streams.append(torch.cuda.Stream(dev))

        def run(n):
            for i in range(n):
                l = i % lanes
                with torch.cuda.stream(streams[l]):
                    models[l](dict(clips[i % len(clips)]))
        run(4 * lanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.calls)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"lanes {lanes}: {1e3 * (t2 - t0) / a.calls:.3f} ms/clip  {a.calls * T / (t2 - t0):.0f} frames/s   "
              f"(host enqueue {1e3 * (t1 - t0) / a.calls:.3f} ms/clip)", flush=True)
        del models, streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
