"""Whole-clip sharding across the GPUs of one node and the RCCL all-gather of predictions.

The reference is single-process ("Parallelism strategies present in the reference:
none"; its batch is one clip, `dataloader/wk_action_genome.py:622-627`).  Clips never interact, so
the MI355X design is: one process per GPU, a full weight replica each, clips assigned statically,
no collective on the data path, and ONE fixed-stride all-gather per batch of clips that brings
every rank's `[pairs, 26]` prediction rows to all ranks (rank 0 then runs the evaluator).
`torch.distributed`'s "nccl" backend is RCCL on ROCm; the same code runs on "gloo" for CPU tests.

Two entry points over the same code:
  * `PredictionGatherer` -- the pipelined form `bench.py --gpus N` times: fixed capacities, a ring of
    buffers, gathers issued asynchronously (RCCL's stream, under the next forward), **no host
    synchronisation** until a result is asked for;
  * `all_gather_predictions` -- one-shot convenience (returns the per-clip dict); without capacities it
    first exchanges sizes, which costs two host synchronisations.
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


def pack_predictions(pred, out=None):
    """[P, 3 + 6 + 17] rows: attention logits | spatial | contacting probabilities (into `out[:P]` if given)."""
    parts = [pred["attention_distribution"], pred["spatial_distribution"], pred["contacting_distribution"]]
    if out is None:
        return torch.cat(parts, dim=1)
    P = parts[0].shape[0]
    torch.cat(parts, dim=1, out=out[:P])
    return out[:P]


class PredictionGatherer:
    """Fixed-capacity, pipelined all-gather of per-clip prediction rows.

    Every rank contributes one `[rows_cap, cols]` payload and one `[clips_cap, 2]` int64 record of
    (global clip id, rows of that clip) per `submit`; unused slots carry id -1.  `depth` buffer sets form
    a ring: `submit` first makes the current stream wait for the gather that used the same set `depth`
    submits ago, copies the rows in and issues both all-gathers with `async_op=True` -- on RCCL that
    only enqueues (the collectives run on RCCL's stream, ordered after the copy by the work handle);
    nothing on this path reads a device value back.  `result(ticket)` is the one synchronising call.
    The clip records are uploaded once per distinct (ids, rows) layout and cached on the device."""

    def __init__(self, rows_cap, clips_cap, cols=26, device=None, group=None, depth=2, dtype=torch.float32):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rows_cap, self.clips_cap, self.cols, self.depth = int(rows_cap), int(clips_cap), int(cols), int(depth)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.device = dev
        self._payload = [torch.zeros((self.rows_cap, cols), device=dev, dtype=dtype) for _ in range(depth)]
        self._gathered = [torch.empty((self.world * self.rows_cap, cols), device=dev, dtype=dtype) for _ in range(depth)]
        self._meta_all = [torch.empty((self.world * self.clips_cap, 2), device=dev, dtype=torch.int64) for _ in range(depth)]
        self._works = [None] * depth
        self._meta_cache = {}
        self._n = 0
        self._overflow_seen = torch.zeros((), device=dev, dtype=torch.int64)   # OVERFLOW records met by `gathered()`

    OVERFLOW = -2          # clip id of a record that says "this rank's submit did not fit its capacities"

    def _meta(self, clip_ids, clip_pairs):
        key = (tuple(int(i) for i in clip_ids), tuple(int(p) for p in clip_pairs))
        m = self._meta_cache.get(key)
        if m is None:
            if len(key[0]) != len(key[1]):
                raise ValueError(f"{len(key[0])} clips but {len(key[1])} row counts")
            host = torch.full((self.clips_cap, 2), -1, dtype=torch.int64)
            if len(key[0]) > self.clips_cap or sum(key[1]) > self.rows_cap:
                # The capacities are a rank-local fact: raising here would leave the OTHER ranks hanging in the
                # all-gather.  The rank still takes part, sending an OVERFLOW record (and no rows); every rank's
                # `result()` then raises the same error.
                host[0, 0], host[0, 1] = self.OVERFLOW, sum(key[1])
            elif key[0]:
                host[: len(key[0]), 0] = torch.tensor(key[0], dtype=torch.int64)
                host[: len(key[0]), 1] = torch.tensor(key[1], dtype=torch.int64)
            m = host.to(self.device)
            if len(self._meta_cache) > 64:
                self._meta_cache.clear()
            self._meta_cache[key] = m
        return m

    def payload(self):
        """The `[rows_cap, cols]` buffer the NEXT submit will send: a producer may write its rows straight into it
        (`pack_predictions(pred, out=g.payload())`) and pass the returned view to `submit` -- no extra copy."""
        k = self._n % self.depth
        self._wait(k)
        return self._payload[k]

    def _wait(self, k):
        if self._works[k] is not None:
            for w in self._works[k]:
                w.wait()
            self._works[k] = None
            # a rank whose submit did not fit sends an OVERFLOW record and no rows: counted HERE, once per gather, on the
            # first wait for its buffer set (ring reuse, `wait_all`, `gathered`, `result`) and where it lives (no read-back)
            # -- so a loop that only submits (bench.py's timed pass) still learns of it through `raise_if_overflowed()`
            meta = self._meta_all[k].view(self.world, self.clips_cap, 2)
            self._overflow_seen += (meta[:, 0, 0] == self.OVERFLOW).sum()

    def submit(self, local_rows, clip_ids, clip_pairs):
        """Issue the gather of this rank's rows (`[sum(clip_pairs), cols]`); returns a ticket for `result`."""
        k = self._n % self.depth
        self._wait(k)
        n = int(local_rows.shape[0])
        meta = self._meta(clip_ids, clip_pairs)
        if n != sum(int(p) for p in clip_pairs):
            raise ValueError("local_rows does not have sum(clip_pairs) rows")
        buf = self._payload[k]
        if n and n <= self.rows_cap and local_rows.data_ptr() != buf.data_ptr():
            buf[:n].copy_(local_rows)
        w1 = dist.all_gather_into_tensor(self._meta_all[k], meta, group=self.group, async_op=True)
        w2 = dist.all_gather_into_tensor(self._gathered[k], buf, group=self.group, async_op=True)
        self._works[k] = (w1, w2)
        ticket = self._n
        self._n += 1
        return ticket

    def wait_all(self):
        """Make the current stream (gloo: the host) wait for every gather still in flight."""
        for k in range(self.depth):
            self._wait(k)

    def gathered(self, ticket):
        """The raw buffers of gather `ticket` WITHOUT a host synchronisation: ([world, rows_cap, cols] rows,
        [world, clips_cap, 2] clip records), ordered after the gather on the current stream (gloo: the host waits).
        For consumers that know every rank's layout already (bench.py's strong-scaling loop: the clip assignment is a
        deterministic function every rank computes) and only need the rows.  Valid until the buffer set is reused."""
        if not (self._n - self.depth <= ticket < self._n):
            raise ValueError("that gather's buffers have been reused")
        k = ticket % self.depth
        self._wait(k)
        meta = self._meta_all[k].view(self.world, self.clips_cap, 2)
        # (OVERFLOW records were counted by `_wait`, once per gather: `raise_if_overflowed()` -- one synchronisation,
        # whenever the consumer likes -- turns them into the error `result()` raises, so a consumer of the raw buffers
        # cannot score stale rows without ever hearing of it, however often it asks for the same ticket)
        return self._gathered[k].view(self.world, self.rows_cap, self.cols), meta

    def raise_if_overflowed(self):
        """Synchronises: raises if any gather waited for so far (`wait_all` first, to cover the ones in flight) carried
        an OVERFLOW record."""
        n = int(self._overflow_seen.item())
        if n:
            self._overflow_seen.zero_()
            raise ValueError(f"{n} rank submit(s) exceeded the gatherer's capacities (rows_cap={self.rows_cap}, "
                             f"clips_cap={self.clips_cap}): the rows handed out by gathered() for them are not predictions")

    def result(self, ticket):
        """{clip_id: [pairs, cols] tensor} of the gather `ticket` (views into its buffer set: valid until that set
        is reused, i.e. for the next `depth - 1` submits).  Synchronises: reads the clip records back."""
        if not (self._n - self.depth <= ticket < self._n):
            raise ValueError("that gather's buffers have been reused")
        k = ticket % self.depth
        self._wait(k)
        meta = self._meta_all[k].view(self.world, self.clips_cap, 2).cpu()
        out = {}
        for r in range(self.world):
            off = r * self.rows_cap
            for cid, rows in meta[r].tolist():
                if cid == self.OVERFLOW:
                    raise ValueError(f"rank {r} submitted {rows} prediction rows / more clips than the gatherer's "
                                     f"capacities (rows_cap={self.rows_cap}, clips_cap={self.clips_cap}) hold")
                if cid < 0:
                    continue
                out[int(cid)] = self._gathered[k][off: off + rows]
                off += rows
        return out


def all_gather_predictions(local_rows, local_clip_ids, local_clip_pairs, group=None, rows_cap=None, clips_cap=None):
    """Gather per-clip prediction rows from every rank (one-shot form of `PredictionGatherer`).

    local_rows       [sum(local_clip_pairs), C] tensor on this rank's device
    local_clip_ids   global ids of this rank's clips, in row order
    local_clip_pairs rows per clip
    rows_cap, clips_cap  known upper bounds per rank of rows / clips.  With BOTH given nothing is exchanged
                     beforehand and the only host synchronisation is the final read of the clip records; if either
                     is missing the ranks first agree on it (one all-reduce + `.item()` each).
    Returns {clip_id: [pairs, C] tensor} for all clips of all ranks (views into one gathered buffer).
    The payload is sub-MB per clip, i.e. latency-bound, so it is sent once per batch, not per clip."""
    dev = local_rows.device
    if rows_cap is None or clips_cap is None:
        need = torch.tensor([len(local_clip_ids), int(local_rows.shape[0])], device=dev, dtype=torch.int64)
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
        n_clips, n_rows = (int(v) for v in need.tolist())
        clips_cap = max(n_clips, 1) if clips_cap is None else clips_cap
        rows_cap = max(n_rows, 1) if rows_cap is None else rows_cap
    g = PredictionGatherer(rows_cap, clips_cap, cols=local_rows.shape[1], device=dev, group=group, depth=1,
                           dtype=local_rows.dtype)
    return g.result(g.submit(local_rows, local_clip_ids, local_clip_pairs))


def all_reduce_recall(evaluator, group=None, device=None):
    """Recall table of a split whose clips were scored by SEVERAL ranks, each with its own evaluator on its own clips:
    one all-reduce (SUM) of `evaluator.partial_sums()` -- 2 * (9 + 6 * num_rel) float64 -- and every rank holds the
    `summary()` of the union.  Replaces "gather every prediction row to rank 0 and score there", which serialises a
    strong-scaling run on rank 0's evaluator; the all-gather of the rows (`PredictionGatherer`) remains the way to bring
    the predictions themselves to every rank.  No reference counterpart (single process, `tools/test_STTran.py:62-92`)."""
    vec = torch.from_numpy(evaluator.partial_sums())
    if dist.is_available() and dist.is_initialized():
        if device is None and dist.get_backend(group) == "nccl":      # RCCL reduces device tensors only
            device = torch.device("cuda", torch.cuda.current_device())
        if device is not None:
            vec = vec.to(device)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)      # (a 1-rank group: the identity, still the collective)
    return evaluator.summary_from_partial_sums(vec.cpu().numpy())
