"""Shared pieces of bench.py's legs: the synthetic clip generator of SURVEY.md 8(d), the per-process environment (one
process per GPU, rendezvous, barrier, max-over-ranks) and the model factory.  Nothing here is timed."""
import json
import os
import socket
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nl_vsgg_amd.lib import synthetic as syn  # noqa: E402
from nl_vsgg_amd.lib.sttran import STTran  # noqa: E402

CLASSES = ["__background__"] + [f"c{i}" for i in range(36)]
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16X3_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS / 6    # the bf16x3 emulation's roof in fp32-equivalent TFLOP/s (419.5)
# frames, boxes per frame, default clips per step.  A step batches ~10 k pairs (64 clips of 16x12 = 11 264 pairs, 4 clips of
# 64x36 = 8 960): the clip is the BASELINE one, the batch is this framework's (`pack_clips`); measured on one MI355X the
# 16x12 rate goes 15.8 k (1 clip) -> 28.3 k (8) -> 30.2 k (16) -> 31.3 k (32) -> 32.0 k (64) -> 32.5 k (128) frames/s as
# tile quantisation and the stream-K fix-ups amortise; `batch_sweep` in the line re-measures 1 / 16 / default every run.
SHAPES = {"16x12": (16, 12, 64), "64x36": (64, 36, 4)}


def device_clip(T, N, gen, device, shifted=False):
    """One synthetic clip of T frames x N boxes (1 person + N-1 objects per frame) built on the
    device with the distributions of SURVEY.md 8(d).  `shifted`: the same totals (T frames, T*N boxes, T*(N-1) pairs,
    the same number of window tokens) with ONE object moved from frame T//4 to frame T//2 -- another per-frame pair-count
    vector, i.e. another layout for the library's index-map cache, at the same work."""
    counts = np.full(T, N - 1, dtype=np.int64)
    if shifted:
        if T < 4 or N < 3:
            raise ValueError("a shifted clip needs >= 4 frames and >= 3 boxes per frame")
        counts[T // 4] -= 1
        counts[T // 2] += 1
    B, P = int(T + counts.sum()), int(counts.sum())
    cd = torch.from_numpy(counts).to(device)
    first = torch.cumsum(cd + 1, 0) - (cd + 1)                                  # the person box of each frame
    fr = torch.arange(T, device=device).repeat_interleave(cd)
    start = torch.cumsum(cd, 0) - cd
    obj = torch.arange(P, device=device) - start[fr] + 1                        # 1 .. pairs of the frame
    labels = torch.randint(2, 37, (B,), device=device, generator=gen)
    labels[first] = 1
    return {
        "features": torch.randn(B, 2048, device=device, generator=gen),
        "union_feat": torch.randn(P, 2048, 7, 7, device=device, generator=gen),
        "spatial_masks": torch.rand(P, 2, 27, 27, device=device, generator=gen) - 0.5,
        "labels": labels,
        "pair_idx": torch.stack([first[fr], first[fr] + obj], dim=1),
        "im_idx": fr.float(),
        "frame_counts": counts.astype(np.int32),
        "num_frames": T,
    }



def make_batch(env, model_kind, T, N, cps, seed, shifted=False):
    """cps clips of T x N; `shifted`: clip 0 carries another per-frame pair-count vector (device_clip) at the same totals"""
    device = env.device
    gen = torch.Generator(device=device).manual_seed(seed + env.rank)
    clips = [device_clip(T, N, gen, device, shifted=shifted and i == 0) for i in range(cps)]
    if model_kind == "dsgdetr":                   # sgdet entry: detector boxes, class distribution, scores
        for c in clips:
            B = c["features"].shape[0]
            xy = torch.rand(B, 2, device=device, generator=gen) * 300
            wh = torch.rand(B, 2, device=device, generator=gen) * 150 + 10
            frame_of_box = torch.arange(T, device=device).repeat_interleave(torch.from_numpy(c["frame_counts"] + 1).to(device))
            c["boxes"] = torch.cat([frame_of_box[:, None].float(), xy, xy + wh], 1)
            c["distribution"] = torch.softmax(torch.randn(B, 36, device=device, generator=gen), 1)
            c["scores"] = c["distribution"].max(1).values
            c["im_idx"] = c["im_idx"].long()
    return clips


def pci_bus_id(ordinal):
    """PCI bus id of a HIP device ordinal ("0000:c5:00.0")."""
    pr = torch.cuda.get_device_properties(ordinal)
    if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if ctypes.CDLL("libamdhip64.so").hipDeviceGetPCIBusId(buf, 64, int(ordinal)) == 0:
            return buf.value.decode().lower()
    except Exception:
        pass
    return None



class Env:
    """process-wide state shared by the workload runs"""
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run")
        # one process per GPU; BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only so the N>1 code path can be
        # smoke-tested on a single-GPU box (all ranks on device 0, gloo instead of RCCL): tests/test_bench_gpu.py
        self.local = int(os.environ.get("BENCH_FORCE_DEVICE", local))
        ndev = torch.cuda.device_count()
        forced = "BENCH_FORCE_DEVICE" in os.environ
        # the ordinals of THIS node's ranks (LOCAL_WORLD_SIZE of them: 8 on each of two nodes is a 16-rank job whose every
        # local ordinal exists); a launcher that narrows each rank to one visible device sets LOCAL_WORLD_SIZE = 1 or
        # gives every rank ordinal 0 through BENCH_FORCE_DEVICE
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
        missing = [r for r in range(local_world) if r >= ndev] if not forced else ([self.local] if self.local >= ndev else [])
        if missing:
            # A mis-provisioned node (fewer GPUs than local ranks; LOCAL_RANK = ordinal) must still leave a parseable
            # record: EVERY rank of the node leaves before the rendezvous (the ranks whose ordinal exists would wait for
            # the others in init_process_group) and rank 0 -- whose ordinal 0 exists whenever any GPU does -- prints a
            # compact line carrying "error"; exit code 2.
            msg = f"rank {missing[0]}: no device {missing[0]} ({ndev} GPU(s) visible, {local_world} local rank(s) of {self.world})"
            print(f"bench.py: rank {self.rank}: {msg}", file=sys.stderr)
            if self.rank == 0:
                from .line import error_line
                print(error_line(args, self.world, msg), flush=True)
            raise SystemExit(2)
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        # host threads of this rank: the evaluator's tally, torch's CPU ops and numpy run in this process next to 7 others
        # on an 8-GPU node -- cap torch's intra-op pool so N ranks do not each start one thread per host core
        cores = os.cpu_count() or 1
        self.host_threads = max(1, cores // (2 * self.world)) if self.world > 1 else None
        if self.host_threads:
            torch.set_num_threads(self.host_threads)
        self.dist = None
        if self.world > 1 or getattr(args, "rccl_selftest", False):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1 and "MASTER_PORT" not in os.environ:      # --rccl-selftest without a launcher
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"))    # nccl == RCCL on ROCm
            self.dist = dist

    def devices(self):
        """what every rank ran on, all-gathered: lets the reader check that the N ranks sat on N distinct GPUs"""
        pr = torch.cuda.get_device_properties(self.local)
        mine = {"rank": self.rank, "device": self.local, "pci_bus_id": pci_bus_id(self.local), "name": pr.name,
                "uuid": str(getattr(pr, "uuid", "")) or None, "pid": os.getpid(),
                "backend": self.dist.get_backend() if self.dist else None}
        if self.dist is None:
            return [mine]
        out = [None] * self.world
        self.dist.all_gather_object(out, mine)
        return out

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self, gatherer=None):
        if self.dist is not None:
            if gatherer is not None:
                gatherer.wait_all()
            if self.dist.get_backend() == "nccl":
                self.dist.barrier(device_ids=[self.local])
            else:
                self.dist.barrier()
        torch.cuda.synchronize()



def flush_c_stdio():
    """RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe or a file and would then
    be written at process exit -- BEHIND the JSON line.  Flush it out before the line is printed, so the line stays last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def make_model(kind, device, gemm_engine="fp32"):
    """(model, state dict) of `kind` ("sttran" = PredCls STTran, "dsgdetr" = lib/dsg_detr.py's sgdet branch) with the seeded
    synthetic weights of lib/synthetic.py, set up the way every leg wants it: enqueue-only, no hidden input copies."""
    if kind == "dsgdetr":
        from nl_vsgg_amd.lib.dsg_detr import STTran as DSGDETR
        sd = syn.make_dsg_detr_state_dict(7)
        model = DSGDETR(mode="sgdet", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                        obj_classes=CLASSES).to(device)
    else:
        sd = syn.make_sttran_state_dict(7)
        model = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                       obj_classes=CLASSES, enc_layer_num=1, dec_layer_num=3, transformer_mode="wk", is_wks=True,
                       feat_dim=2048).to(device)
    model.eval()
    model.check_indices = False      # enqueue-only: no per-call synchronisation inside the timed region
    model.strict_inputs = True       # a hidden per-step copy of the inputs would be timed as compute
    model.gemm_engine = gemm_engine
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    return model, sd
