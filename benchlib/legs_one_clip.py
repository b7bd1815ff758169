"""The reference's own batch size -- ONE clip per `model(entry)` call (dataloader/wk_action_genome.py:622-627, loop body
tools/test_STTran.py:75-88) -- in the forms a user of the shim can run it, plus the batch sizes between one clip and the
default.  None of these is the line's `value`."""
import collections
import time

import torch

from nl_vsgg_amd.lib.sttran import pack_clips

# calls in flight in the one-clip-per-pass leg.  3 lanes + the caller's stream = the 4 hardware queues a HIP process gets by
# default (GPU_MAX_HW_QUEUES): measured 2 / 3 / 4 / 6 / 8 lanes = 18.8 / 20.9 / 18.8 / 18.5 / 20.3 k frames/s on a box whose
# serial rate was 14.4 k (tools/experiments/lanes_probe.py --api) -- more lanes than queues share queues again
ONE_CLIP_LANES = {"16x12": 3, "64x36": 2}        # (64x36: 2 lanes 9.88-9.94 k, 4 lanes 9.64-10.0 k, serial 9.26-9.34 k frames/s)
# entries per coalesced group (`model.coalesce`): the reference's loop body unchanged, K calls issued as one by-pointer
# forward on the next lane
ONE_CLIP_COALESCE = {"16x12": 16, "64x36": 4}
SWEEP_CPS = {"16x12": 16, "64x36": 1}                           # the smaller batch of `batch_sweep` (round 1-2 defaults)


def _timed(fn, n, warm=8):
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def one_clip_legs(w):
    """Two clips alternate (clip 0 of each batch: other tensors, other per-frame counts): every call is a new entry, as in
    the reference's loop.
      `one_clip_per_pass.serial`  `model(entry)` on the caller's stream, one clip at a time: INTEGRATION.md's one-import
                                  switch with the loop untouched (rounds 1-3's figure);
      `one_clip_per_pass.value`   the same calls as `model.forward_async(entry)` with ONE_CLIP_LANES lanes in the handle:
                                  call i runs on lane i % lanes' own stream and is joined (event wait, no host
                                  synchronisation) lanes - 1 calls later, as a pipelined consumer would -- the meaning this key
                                  had in rounds 1-4 (round 5 had put the coalesced figure here);
      `one_clip_coalesced.value`  the same loop body with `model.coalesce = K`: every K calls are issued as ONE by-pointer
                                  forward on the next lane and each entry gets its rows as views; the caller keeps
                                  `model.pipeline_depth` (= lanes x K) entries un-joined, so the result of a call is available
                                  `result_latency_ms` after it was submitted (measured below), not one forward later."""
    model, T = w.model, w.T
    ones = [b[0] for b in w.batches]
    n1 = 2 * max(w.steps, 10)

    def loop_serial(n):
        for i in range(n):
            model(dict(ones[i % len(ones)]))

    def loop_lanes(n):
        pending = collections.deque()
        for i in range(n):
            pending.append(model.forward_async(dict(ones[i % len(ones)])))
            if len(pending) == model.lanes:
                model.join(pending.popleft())
        while pending:
            model.join(pending.popleft())

    dt_serial = _timed(loop_serial, n1)
    nlanes = ONE_CLIP_LANES[w.name]
    model.lanes = nlanes
    model.reserve(int(ones[0]["pair_idx"].shape[0]) + 8, int(ones[0]["features"].shape[0]) + 8)
    dt1 = _timed(loop_lanes, 2 * n1)
    model.sync_check()
    K = ONE_CLIP_COALESCE[w.name]
    model.coalesce = K
    model.reserve(K * int(ones[0]["pair_idx"].shape[0]) + 8, K * int(ones[0]["features"].shape[0]) + 8)

    def loop_coalesced(n, hints=True, stamps=None):
        pending = collections.deque()

        def pop():
            e, t_sub = pending.popleft()
            model.join(e)
            if stamps is not None:                     # latency probe only: wait until the rows are really there
                torch.cuda.current_stream().synchronize()
                stamps.append(time.perf_counter() - t_sub)
        for i in range(n):
            e = dict(ones[i % len(ones)])
            if not hints:                              # the reference's entry: no host-side frame counts
                e.pop("frame_counts"); e.pop("num_frames")
            pending.append((model.forward_async(e), time.perf_counter()))
            if len(pending) == model.pipeline_depth:
                pop()
        while pending:
            pop()
    n_co = K * nlanes * max(4, min(w.steps, 20) // 2)
    dt_co = _timed(loop_coalesced, n_co, warm=2 * K * nlanes)
    dt_nh = _timed(lambda n: loop_coalesced(n, hints=False), n_co, warm=K * nlanes)
    stamps = []
    loop_coalesced(2 * K * nlanes, stamps=stamps)          # submit -> rows visible to the host, per entry (its own short run)
    torch.cuda.synchronize()
    model.sync_check()
    model.coalesce = 0
    model.lanes = 1
    return {
        "one_clip_per_pass": {"value": T / dt1, "unit": "frames/s", "ms_per_step": 1e3 * dt1, "calls": 2 * n1, "lanes": nlanes,
                              "serial": {"value": T / dt_serial, "ms_per_step": 1e3 * dt_serial},
                              "note": "same clip shape, ONE clip per call (the reference's batch, "
                                      "dataloader/wk_action_genome.py:622-627), a different entry on every call; value = "
                                      f"`model.forward_async(entry)` on {nlanes} lanes, joined lanes - 1 calls later; `serial` = "
                                      "`model(entry)` one call at a time on the caller's stream (the one-import switch)"},
        "one_clip_coalesced": {"value": T / dt_co, "unit": "frames/s", "ms_per_step": 1e3 * dt_co, "calls": n_co,
                               "lanes": nlanes, "coalesce": K, "pipeline_depth": K * nlanes,
                               "no_hints": {"value": T / dt_nh, "ms_per_step": 1e3 * dt_nh},
                               "result_latency_ms": 1e3 * sum(stamps) / max(len(stamps), 1),
                               "note": "the reference's loop body (tools/test_STTran.py:75-88) as "
                                       "`pending.append(model.forward_async(entry))` / `model.join(pred)` with "
                                       f"model.coalesce = {K}: every {K} calls are issued as one by-pointer forward on one of "
                                       f"{nlanes} lanes; `no_hints` = entries without host-side frame_counts (one im_idx "
                                       "read-back per group); result_latency_ms = submit -> rows visible to the host, mean "
                                       "over a separate run that synchronises after every join (the price of the batching)"},
    }


def two_steps_in_flight(w):
    """the headline's batches, two steps in flight on two lanes of the handle (not `value`: steps overlap)"""
    model = w.model
    model.lanes = 2
    model.reserve(w.P, w.B)

    def loop2(n):
        pending = collections.deque()
        for i in range(n):
            pending.append(model.forward_async(pack_clips(w.batches[i % len(w.batches)], copy=False)))
            if len(pending) == 2:
                model.join(pending.popleft())
        while pending:
            model.join(pending.popleft())
    n3 = 2 * max(3, min(w.steps, 12) // 2)
    dt3 = _timed(loop2, n3, warm=4)
    model.sync_check()
    model.lanes = 1
    return {"value": w.frames_per_step / dt3, "ms_per_step": 1e3 * dt3, "lanes": 2, "steps": n3,
            "note": "the same batches with two forwards in flight on two lanes of the handle: the short kernels and tails of "
                    "one step run under the other step's GEMMs"}


def batch_sweep(w):
    """the batch size between one clip and the default (the default of rounds 1-2): a few steps"""
    c2 = SWEEP_CPS[w.name]
    if not w.cps > c2 > 1:
        return None
    model, batches = w.model, w.batches
    for _ in range(2):
        for b in batches:
            model(pack_clips(b[:c2], copy=False))
    torch.cuda.synchronize()
    n2 = 2 * max(3, min(w.steps, 20) // 2)
    t0 = time.perf_counter()
    for i in range(n2):
        model(pack_clips(batches[i % len(batches)][:c2], copy=False))
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n2
    return [{"clips_per_step": c2, "value": c2 * w.T / dt2, "ms_per_step": 1e3 * dt2}]
