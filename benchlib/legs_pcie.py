"""PCIe-inclusive rate (never `value`): every step's inputs start in pinned host memory."""
import time

import torch

from nl_vsgg_amd.lib.sttran import pack_clips


def pcie_legs(w, pcie):
    """`pcie` = "overlapped": the copy of step i+1 on a second stream under the forward of step i (two buffer sets);
    "full": additionally the serial form (H2D, then forward, one stream)."""
    model, device, clips, cps = w.model, w.env.device, w.clips, w.cps
    out = {}
    batch = pack_clips(clips) if cps > 1 else clips[0]          # one contiguous staging area per tensor
    host = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            host[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
            host[k].copy_(v)
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    if pcie == "full":
        def step_h2d():
            b = dict(batch)
            for k, v in host.items():
                b[k] = v.to(device, non_blocking=True)
            return model(b)
        for _ in range(2):
            step_h2d()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(w.steps // 2, 3)):
            step_h2d()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(w.steps // 2, 3)
        out["pcie_inclusive"] = {"value": w.frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                 "h2d_bytes_per_step": nbytes,
                                 "note": "serial H2D (pinned) + forward on one stream, no overlap"}
    # the same with the copy of step i+1 on a second stream under the forward of step i (two buffer sets)
    copy_stream, main = torch.cuda.Stream(device), torch.cuda.current_stream(device)
    bufs = [{k: torch.empty_like(batch[k]) for k in host} for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]       # buffer filled
    freed = [torch.cuda.Event() for _ in range(2)]       # forward that read the buffer has finished

    def upload(slot):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[slot])
            for k, v in host.items():
                bufs[slot][k].copy_(v, non_blocking=True)
            ready[slot].record(copy_stream)

    def pipelined(n):
        for e in freed:
            e.record(main)
        upload(0)
        for i in range(n):
            slot = i & 1
            if i + 1 < n:
                upload(slot ^ 1)
            main.wait_event(ready[slot])
            b = dict(batch); b.update(bufs[slot])
            model(b)
            freed[slot].record(main)
    pipelined(3)
    torch.cuda.synchronize()
    n_over = max(w.steps, 6) if pcie == "full" else 6
    t0 = time.perf_counter()
    pipelined(n_over)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_over
    out["pcie_inclusive_overlapped"] = {"value": w.frames_per_step / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt,
                                        "h2d_bytes_per_step": nbytes, "h2d_gb_per_s": nbytes / dt / 1e9, "steps": n_over,
                                        "note": "inputs start in pinned host memory on every step: H2D of step i+1 on a copy "
                                                "stream under the forward of step i (never `value`)"}
    del host, bufs, batch
    return out
