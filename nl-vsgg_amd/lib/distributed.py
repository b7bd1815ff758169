"""Whole-clip sharding across the GPUs of one node and the RCCL all-gather of predictions.

The reference is single-process ("Parallelism strategies present in the reference:
none"; its batch is one clip, `dataloader/wk_action_genome.py:622-627`).  Clips never interact, so
the MI355X design is: one process per GPU, a full weight replica each, clips assigned statically,
no collective on the data path, and ONE fixed-stride all-gather per batch of clips that brings
every rank's `[pairs, 26]` prediction rows to all ranks (rank 0 then runs the evaluator).
`torch.distributed`'s "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def assign_clips(costs, world_size):
    """Longest-processing-time-first assignment of clips to ranks.

    costs[i] ~ work of clip i (pairs x frames is a good proxy: the decoder dominates and is linear in
    both).  Returns `owner[i]`; deterministic, so every rank computes the same map without talking."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        owner[i] = r
        load[r] += float(costs[i])
    return owner


def pack_predictions(pred):
    """[P, 3 + 6 + 17] rows: attention logits | spatial | contacting probabilities."""
    return torch.cat([pred["attention_distribution"], pred["spatial_distribution"],
                      pred["contacting_distribution"]], dim=1)


def all_gather_predictions(local_rows, local_clip_ids, local_clip_pairs, group=None, rows_cap=None):
    """Gather per-clip prediction rows from every rank.

    local_rows      [sum(local_clip_pairs), C] tensor on this rank's device
    local_clip_ids  global ids of this rank's clips, in row order
    local_clip_pairs rows per clip
    rows_cap        optional known upper bound of rows per rank (skips the size exchange; the bench
                    uses it because every rank runs identical shapes)
    Returns {clip_id: [pairs, C] tensor} for all clips of all ranks (views into one gathered buffer).
    One small all-gather of (id, rows) metadata + one all-gather of a fixed-stride payload; the
    payload is sub-MB per clip, i.e. latency-bound, so it is sent once per batch, not per clip."""
    world = dist.get_world_size(group)
    dev, C = local_rows.device, local_rows.shape[1]
    n_local = len(local_clip_ids)
    n_max = torch.tensor([n_local], device=dev, dtype=torch.int64)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX, group=group)
    n_max = int(n_max.item())
    meta = torch.full((n_max, 2), -1, device=dev, dtype=torch.int64)
    if n_local:
        meta[:n_local, 0] = torch.as_tensor(local_clip_ids, device=dev, dtype=torch.int64)
        meta[:n_local, 1] = torch.as_tensor(local_clip_pairs, device=dev, dtype=torch.int64)
    all_meta = torch.empty((world, n_max, 2), device=dev, dtype=torch.int64)
    dist.all_gather_into_tensor(all_meta.view(world * n_max, 2), meta, group=group)
    all_meta = all_meta.cpu()
    if rows_cap is None:
        rows_cap = int(all_meta[:, :, 1].clamp(min=0).sum(dim=1).max().item())
    payload = torch.zeros((rows_cap, C), device=dev, dtype=local_rows.dtype)
    payload[: local_rows.shape[0]] = local_rows
    gathered = torch.empty((world * rows_cap, C), device=dev, dtype=local_rows.dtype)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    out = {}
    for r in range(world):
        off = r * rows_cap
        for cid, rows in all_meta[r].tolist():
            if cid < 0:
                continue
            out[int(cid)] = gathered[off: off + rows]
            off += rows
    return out
