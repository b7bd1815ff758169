#!/usr/bin/env python3
"""Evaluator throughput: host numpy evaluator vs the device evaluator, same clips.

    python tools/eval_bench.py [--workload 16x12|64x36] [--clips 32]

Prints one JSON line: frames/s of both evaluators (ground truth packed once for the device one, as a
real run would do across epochs), the kernel's average duration, and that both give identical recall."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator  # noqa: E402
from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP  # noqa: E402

OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]; SPA = [f"spa{i}" for i in range(6)]; CON = [f"con{i}" for i in range(17)]
KW = dict(AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON, AG_attention_predicates=ATT,
          AG_spatial_predicates=SPA, AG_contacting_predicates=CON, iou_threshold=0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="16x12")
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--host-clips", type=int, default=0, help="clips timed on the host evaluator (default: enough for ~5 s)")
    a = ap.parse_args()
    T, N = (int(x) for x in a.workload.split("x"))
    counts = [N - 1] * T
    clips = []
    for c in range(a.clips):
        e = syn.make_entry(900 + c, counts, geometry_only=True)
        gt = syn.make_gt_annotation(1900 + c, e)
        rng = np.random.default_rng(c)
        P = sum(counts)
        pred = {k: e[k] for k in ("pair_idx", "im_idx", "boxes", "labels", "scores")}
        pred["attention_distribution"] = (3 * rng.standard_normal((P, 3))).astype(np.float32)
        pred["spatial_distribution"] = rng.random((P, 6)).astype(np.float32)
        pred["contacting_distribution"] = rng.random((P, 17)).astype(np.float32)
        clips.append((gt, pred))
    host = SceneGraphEvaluator(mode="predcls", **KW); host.register_container()
    dev = SceneGraphEvaluator_HIP(mode="predcls", **KW); dev.register_container()
    dpred = [{k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in p.items()} for _, p in clips]

    t0 = time.perf_counter()
    packed = [dev.pack(gt) for gt, _ in clips]
    t_pack = time.perf_counter() - t0
    for pk in packed:
        pk.on(torch.device("cuda", 0))
    gc.collect(); gc.freeze()          # the synthetic clips are millions of small objects: keep the collector off them
    # warm-up, then the timed pass: enqueue all clips, one flush
    for pk, p in zip(packed, dpred):       # one full untimed pass (numpy / torch lazy initialisation)
        dev.evaluate_scene_graph(pk, p)
    dev.flush()
    dev.register_container()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for pk, p in zip(packed, dpred):
        dev.evaluate_scene_graph(pk, p)
    dev.flush()
    t_dev = time.perf_counter() - t0
    # kernel time alone
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scratch = SceneGraphEvaluator_HIP(mode="predcls", **KW); scratch.register_container()
    ev0.record()
    for pk, p in zip(packed, dpred):
        scratch.evaluate_scene_graph(pk, p)
    ev1.record(); torch.cuda.synchronize()
    kernel_us = 1e3 * ev0.elapsed_time(ev1) / len(clips)

    n_host = a.host_clips or len(clips)
    t0 = time.perf_counter()
    done = 0
    for gt, p in clips[:n_host]:
        host.evaluate_scene_graph(gt, p); done += 1
        if not a.host_clips and time.perf_counter() - t0 > 5.0:
            break
    t_host = time.perf_counter() - t0
    same = None
    if done == len(clips):
        host.calculate_mean_recall(); dev.calculate_mean_recall()
        same = host.summary() == dev.summary()
    print(json.dumps({
        "workload": a.workload, "clips": len(clips), "frames_per_clip": T, "pairs_per_clip": sum(counts),
        "device_evaluator_frames_per_s": len(clips) * T / t_dev,
        "device_kernel_us_per_clip": kernel_us,
        "gt_pack_ms_per_clip_once": 1e3 * t_pack / len(clips),
        "host_evaluator_frames_per_s": done * T / t_host, "host_clips_timed": done,
        "identical_summary": same}))


if __name__ == "__main__":
    main()
