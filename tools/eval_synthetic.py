#!/usr/bin/env python3
"""Counterpart of the reference's inference driver (`tools/test_STTran.py:31-92`) on synthetic,
Action-Genome-shaped clips: per clip `pred = model(entry)` -> `evaluator.evaluate_scene_graph(gt, pred)`,
then `print_stats`.  Real AG annotations / frames / VinVL features / checkpoints are not shipped with
the reference, so clips are synthetic: the number of frames per clip follows the AG
test split's statistics (mean 31, capped), 1-6 pairs per frame, seeded weights.

    python tools/eval_synthetic.py --clips 32                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 tools/eval_synthetic.py --clips 256

The loop keeps `--lanes` clips in flight on the handle's lanes (`model.forward_async` / `model.join`):
the reference forwards one clip per call, and one clip cannot fill an MI355X.

With several ranks, clips are assigned by `assign_clips` (longest-processing-time-first) and every rank runs its share.
`--merge tallies` (default): every rank also SCORES its own clips and the recall tallies are merged with one all-reduce
(`all_reduce_recall`); `--merge gather`: predictions are exchanged with ONE RCCL all-gather and rank 0 evaluates all clips."""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.distributed import all_gather_predictions, all_reduce_recall, assign_clips, pack_predictions  # noqa: E402
from nl_vsgg_amd.lib.evaluation_recall import SceneGraphEvaluator  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran  # noqa: E402

OBJ = ["__background__"] + [f"c{i}" for i in range(36)]
ATT = [f"att{i}" for i in range(3)]; SPA = [f"spa{i}" for i in range(6)]; CON = [f"con{i}" for i in range(17)]


def clip_shape(i, seed):
    st = syn.Stream(seed, f"clip{i}")
    T = int(np.clip(3 + st.randint(0, 56, 1)[0] // 1, 3, 60))           # AG test split: 3..121 frames, mean 31
    return [int(c) for c in st.randint(1, 6, T)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=16)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--evaluator", choices=("hip", "host"), default="hip",
                    help="hip: per-frame matching on the GPU (sttran_eval_recall); host: the numpy evaluator")
    ap.add_argument("--lanes", type=int, default=3, help="clips in flight on the handle's lanes (1 = the serial loop)")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="model.coalesce: every K one-clip calls are issued as one by-pointer forward on a lane (0 = off)")
    ap.add_argument("--merge", choices=("tallies", "gather"), default="tallies")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    shapes = [clip_shape(i, a.seed) for i in range(a.clips)]
    owner = assign_clips([sum(s) * len(s) for s in shapes], world)
    mine = [i for i in range(a.clips) if owner[i] == rank]
    model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17, obj_classes=OBJ,
                   enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True, feat_dim=2048).to(dev)
    model.eval()
    model.check_indices = False      # throughput loop: enqueue only (the default, True, synchronises every call)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_sttran_state_dict(7).items()},
                          strict=False)
    entries = {i: syn.make_entry(a.seed + i, shapes[i], real_masks=True) for i in mine}
    kw = dict(mode="predcls", AG_object_classes=OBJ, AG_all_predicates=ATT + SPA + CON,
              AG_attention_predicates=ATT, AG_spatial_predicates=SPA, AG_contacting_predicates=CON,
              iou_threshold=0.5)

    def evaluator():
        if a.evaluator == "hip":
            from nl_vsgg_amd.lib.evaluation_recall_hip import SceneGraphEvaluator_HIP
            ev = SceneGraphEvaluator_HIP(**kw)
        else:
            ev = SceneGraphEvaluator(**kw)
        ev.register_container()
        return ev
    own = evaluator() if a.merge == "tallies" or world == 1 else None
    model.lanes = a.lanes
    model.coalesce = a.coalesce
    depth = model.pipeline_depth if a.coalesce > 1 else a.lanes
    rows, t0, frames = {}, time.perf_counter(), 0
    pending = collections.deque()

    def consume(i, pred):
        pred = model.join(pred)                          # the current stream waits for the lane that computed this clip
        rows[i] = pack_predictions(pred)
        if own is not None:                              # the reference's loop: evaluate right behind the forward
            p = pred if a.evaluator == "hip" else {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in pred.items()}
            own.evaluate_scene_graph(syn.make_gt_annotation(10_000 + a.seed + i, entries[i]), p)
    with torch.no_grad():
        for i in mine:
            e = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) and k != "frame_counts" else v)
                 for k, v in entries[i].items()}
            pending.append((i, model.forward_async(e) if a.lanes > 1 or a.coalesce > 1 else model(e)))
            frames += len(shapes[i])
            if len(pending) >= depth:
                consume(*pending.popleft())
        while pending:
            consume(*pending.popleft())
    model.sync_check()
    dt = time.perf_counter() - t0
    print(f"[rank {rank}] {len(mine)} clips, {frames} frames in {dt:.3f} s (incl. H2D of the synthetic features"
          + (", ground truth and scoring)" if own is not None else ")"))
    if own is not None:
        own.calculate_mean_recall()
        table = all_reduce_recall(own, device=dev)       # one all-reduce of (sum, count) tallies; a no-op on one rank
        if rank == 0:
            if world == 1:
                own.print_stats()
            else:
                for t, name in (("recall", "R"), ("recall_nogc", "R (no constraint)"), ("semi_recall", "R (semi)"),
                                ("mean_recall", "mR"), ("ng_mean_recall", "ng-mR")):
                    print(f"SGG eval ({world} ranks merged): " + "".join(f"{name} @ {k}: {table[t][k]:.4f}; " for k in (10, 20, 50)))
    else:
        local_rows = torch.cat([rows[i] for i in mine]) if mine else torch.zeros((0, 26), device=dev)
        got = all_gather_predictions(local_rows, mine, [sum(shapes[i]) for i in mine])
        if rank == 0:
            ev = evaluator()
            t0 = time.perf_counter()
            for i in range(a.clips):
                e = entries.get(i) or syn.make_entry(a.seed + i, shapes[i], geometry_only=True)
                p = got[i] if a.evaluator == "hip" else got[i].cpu()
                pred = {"attention_distribution": p[:, :3], "spatial_distribution": p[:, 3:9],
                        "contacting_distribution": p[:, 9:], "pair_idx": e["pair_idx"], "im_idx": e["im_idx"],
                        "boxes": e["boxes"], "labels": e["labels"], "scores": e["scores"]}
                ev.evaluate_scene_graph(syn.make_gt_annotation(10_000 + a.seed + i, e), pred)
            ev.calculate_mean_recall()
            print(f"[rank 0] {a.evaluator} evaluator: {a.clips} clips in {time.perf_counter() - t0:.3f} s "
                  "(incl. building the synthetic ground truth)")
            ev.print_stats()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
