"""Host-side mirror of the reference `lib/sttran.py::STTran` on top of the HIP library.

Same constructor signature (`lib/sttran.py:316-318`), same call protocol as
`tools/test_STTran.py:38-52,84`:

    model = STTran(mode=..., attention_class_num=3, spatial_class_num=6, contact_class_num=17,
                   obj_classes=classes, enc_layer_num=1, dec_layer_num=3, transformer_mode='wk',
                   is_wks=True, feat_dim=2048).to(device)
    model.eval(); model.load_state_dict(ckpt['state_dict'], strict=False)
    pred = model(entry)            # mutates and returns `entry`

`entry` is the dict of CUDA tensors the reference detector produces; the keys
written are the ones `lib/evaluation_recall.py:397-465` reads.  PyTorch is used only for device
memory and the current stream; all compute happens in `csrc/` through the C ABI.  There is no
CPU path: without the HIP library (or without a GPU) construction of the handle raises.
"""
from __future__ import annotations

import atexit
import collections
import ctypes as C
import sys
import warnings
import weakref

import numpy as np
import torch

from .. import _native as nat

_IncompatibleKeys = collections.namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])


def _host_i32(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.int32)


# Every model that owns a native handle; destroyed at interpreter exit BEFORE modules (and with them the HIP runtime's
# Python-side owners) are torn down.  atexit handlers run last-registered first: this module is imported after torch.
_LIVE = weakref.WeakSet()


def _destroy_live_handles():
    for m in list(_LIVE):
        try:
            m._destroy()
        except Exception:
            pass


atexit.register(_destroy_live_handles)


class STTran(torch.nn.Module):
    """Inference-only STTran (PredCls, and SGDet with `is_wks=True`).

    An `nn.Module` like the reference's class (lib/sttran.py:316), so `isinstance(model, nn.Module)`, `model.eval()`,
    `.to(device=...)`, `load_state_dict(..., strict=False)`, `state_dict()`, `named_parameters()`, `requires_grad_()` and
    forward hooks behave the way callers such as lib/ults/init_teacher_model.py:15-29 and tools/test_STTran.py:38-52
    expect -- but it registers NO torch parameters: the weights live in the native handle, `state_dict()` returns the
    tensors that were loaded, and `parameters()` is empty (there is nothing to train or to move with `_apply`)."""

    _model = nat.MODEL_STTRAN

    def __init__(self, mode="sgdet", attention_class_num=None, spatial_class_num=None, contact_class_num=None,
                 obj_classes=None, enc_layer_num=None, dec_layer_num=None, transformer_mode=None, is_wks=True,
                 feat_dim=2048, motifs_path=None, conf=None):
        super().__init__()
        assert mode in ("sgdet", "sgcls", "predcls")          # lib/sttran.py:329
        if mode == "sgcls":
            raise NotImplementedError("sgcls is not on the hot path (NL-VSGG runs predcls and sgdet only)")
        # sgdet WITHOUT weak supervision (lib/sttran.py:185-283): the ObjectClassifier selects boxes and
        # pairs (clean_class, per-class NMS, ROIAlign of entry['fmaps']) with lib/object_classifier.py::sgdet_select
        # and computes nothing learnt, so the relation path behind it is the predcls one fed with `pred_labels`.
        self._select = mode == "sgdet" and not is_wks
        self.conf = conf
        self.mode = mode
        self.is_wks = is_wks
        self.obj_classes = obj_classes
        self.attention_class_num = attention_class_num
        self.spatial_class_num = spatial_class_num
        self.contact_class_num = contact_class_num
        self.transformer_mode = transformer_mode
        self.motifs_path = motifs_path
        self.enc_layer_num = int(enc_layer_num)
        self.dec_layer_num = int(dec_layer_num)
        self.feat_dim = int(feat_dim)
        self.training = False           # inference only: constructed in eval mode (`train(True)` raises)
        self.taps = False               # parity tests: also return stage tensors
        # check_indices (default True = the reference's behaviour): out-of-range `pair_idx` / `labels` raise an
        # IndexError like the torch indexing at lib/sttran.py:381-393 does.  The kernels clamp and set a device flag;
        # reading it costs one stream synchronisation per call, which the reference's own loop pays anyway (it
        # synchronises several times per frame, lib/transformer.py:138-140).  Throughput pipelines that only enqueue
        # (bench.py, tools/ag_split_bench.py, HIP-graph capture) set it to False and call `sync_check()` when they like.
        self.check_indices = True
        # strict_inputs: raise instead of silently converting an input that is not already a contiguous tensor of
        # the expected dtype on this model's device (torch would accept it; here a conversion is a hidden copy --
        # 0.9 GB for `union_feat` at 64x36 -- on every call).  Off: convert, and warn once per key for copies > 16 MB.
        self.strict_inputs = False
        self._warned = set()
        # GEMM engine of the nn.Linear layers: "fp32" (default: exact fp32 MFMA) or "bf16x3" (EXPERIMENT: fp32 emulated
        # on the bf16 matrix pipe with three bf16 planes per operand -- fp32-level error, ~1.4x the GEMM rate);
        # "bf16x3_all" = the emulation for every contraction whatever its row count (parity tests on small fixtures)
        self.gemm_engine = "fp32"
        self._engine_set = None
        # lanes (include/sttran_hip.h "LANES"): `model.lanes = K` gives the handle K independent workspaces + streams;
        # `forward_async(entry)` then runs call i on lane i % K without making the current stream wait, so the one-clip
        # calls of the reference's loop overlap on the device; `join(entry)` orders a consumer behind the result.
        self._lanes = 1
        self._lanes_set = 1
        self._next_lane = 0
        self._inflight = {}             # lane -> (_Group, tensors) of its last un-joined call (see _run)
        self._epoch = 0                 # bumped by every device synchronisation (sync_check, set_lanes): older groups are done
        # coalescing (no reference counterpart; the reference's loop is one clip per call, tools/test_STTran.py:75-88 with
        # dataloader/wk_action_genome.py:622-627): `model.coalesce = K` makes `forward_async(entry)` only QUEUE the entry;
        # every K queued entries (or `coalesce_max_pairs` queued pairs, or a `join` / `sync_check` / classic `forward`
        # that needs one of them) are issued as ONE by-pointer batch forward on the next lane -- the unmodified
        # one-clip-per-call loop then runs at the batch path's rate.  See `forward_async`.
        self.coalesce = 0
        self.coalesce_max_pairs = 0     # 0 = no bound; else a group is issued as soon as it holds >= this many pairs
        self._pending = []              # entries queued by forward_async, not yet issued
        self._pending_pairs = 0
        self._device = None
        self._handle = None
        self._sd = collections.OrderedDict()
        self._lib = nat.load()          # raises if the HIP extension is missing

    # ---- the nn.Module surface that differs from the default ------------------------------------------
    def to(self, *args, **kwargs):
        """`.to(device)` / `.to(device=gpu_device)` (tools/test_STTran.py:49, lib/ults/init_teacher_model.py:26): selects
        the MI355X the handle lives on.  dtype arguments are ignored (the path computes in fp32)."""
        device = torch._C._nn._parse_to(*args, **kwargs)[0]
        if device is None:
            return self
        if device.type != "cuda":
            raise RuntimeError("STTran runs on an MI355X only (no CPU path)")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is not None and idx != self._device:
            self._destroy()
        self._device = idx
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def cpu(self):
        raise RuntimeError("STTran runs on an MI355X only (no CPU path)")

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference only (SURVEY.md 2, row 18: training is out of scope)")
        return super().train(False)             # `eval()` is nn.Module's: train(False)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """What was loaded, under the reference's keys (`load_state_dict` keeps the caller's tensors; the device copies
        belong to the native handle)."""
        out = collections.OrderedDict() if destination is None else destination
        for k, v in self._sd.items():
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            out[prefix + k] = t if keep_vars else t.detach()
        return out

    def load_state_dict(self, state_dict, strict=False, assign=False):
        """`model.load_state_dict(ckpt['state_dict'], strict=False)` (tools/test_STTran.py:51-52)."""
        self._sd = collections.OrderedDict(state_dict)
        if self._handle is not None:
            self._upload()
        missing, unexpected = self._key_report()
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing}, unexpected {unexpected}")
        return _IncompatibleKeys(missing, unexpected)

    # ---- handle management -------------------------------------------------------------------
    def _ensure_handle(self):
        if self._handle is not None:
            return
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the STTran hot path has no CPU fallback")
        if self._device is None:
            self._device = torch.cuda.current_device()
        cfg = nat.SttranConfig(
            struct_size=C.sizeof(nat.SttranConfig), device=self._device,
            mode=nat.MODE["predcls" if self._select else self.mode],
            enc_layers=self.enc_layer_num, dec_layers=self.dec_layer_num,
            attention_classes=int(self.attention_class_num), spatial_classes=int(self.spatial_class_num),
            contact_classes=int(self.contact_class_num), num_obj_classes=len(self.obj_classes),
            feat_dim=self.feat_dim, embed_dim=1936, nhead=8, ffn_dim=2048, model=self._model)
        h = C.c_void_p()
        rc = self._lib.sttran_create(C.byref(cfg), C.byref(h))
        if rc != nat.STTRAN_OK:
            raise nat.SttranError(rc, "sttran_create failed")
        self._handle = h
        _LIVE.add(self)
        if self._sd:
            self._upload()

    def _upload(self):
        lib, h = self._lib, self._handle
        for k, v in self._sd.items():
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if not t.dtype.is_floating_point:
                continue                                   # num_batches_tracked
            t = t.detach().to(torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            on_dev = 1 if t.is_cuda else 0
            if t.is_cuda and t.device.index != self._device:
                t, on_dev = t.cpu(), 0
            nat.check(lib, h, lib.sttran_load_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(),
                                                     nat.DTYPE_F32, on_dev))
        if self._key_report()[0] == []:
            nat.check(lib, h, lib.sttran_finalize_weights(h))

    def _key_report(self):
        if self._handle is None:
            return [], []
        buf = C.create_string_buffer(1 << 16)
        n = self._lib.sttran_missing_keys(self._handle, buf, len(buf))
        missing = [s for s in buf.value.decode().split("\n") if s] if n else []
        return missing, []

    def _destroy(self):
        if self._handle is not None:
            self._lib.sttran_destroy(self._handle)
            self._handle = None
            self._engine_set = None
            self._lanes_set = 1
            self._drop_inflight(done=True)
            self._pending, self._pending_pairs = [], 0

    def __del__(self):
        # Never during interpreter shutdown: module teardown order is arbitrary and the HIP runtime underneath
        # sttran_destroy may already be gone (an intermittent abort AFTER a green test run).  Handles that are still
        # alive then were destroyed by the atexit hook below, while everything was intact.
        if sys.is_finalizing():
            return
        try:
            self._destroy()
        except Exception:
            pass

    # ---- profiling hooks used by bench.py ----------------------------------------------------------
    def profile(self, enable=True, reset=True):
        self._ensure_handle()
        if reset:
            nat.check(self._lib, self._handle, self._lib.sttran_profile_reset(self._handle))
        nat.check(self._lib, self._handle, self._lib.sttran_profile_enable(self._handle, 1 if enable else 0))

    def profile_read(self):
        p = nat.SttranProfile(struct_size=C.sizeof(nat.SttranProfile))
        nat.check(self._lib, self._handle, self._lib.sttran_profile_read(self._handle, C.byref(p)))
        return {nat.PROF_NAMES[i]: {"ms": p.ms[i], "flops": p.flops[i], "bytes": p.bytes[i],
                                    "launches": int(p.launches[i])} for i in range(7)} | {"forwards": int(p.forwards)}

    def profile_entries(self):
        """Per (kernel template, shape) breakdown of the same measurements (call after `profile_read`)."""
        n = C.c_int32(0)
        nat.check(self._lib, self._handle, self._lib.sttran_profile_entries(self._handle, None, 0, C.byref(n)))
        arr = (nat.SttranProfEntry * max(n.value, 1))()
        nat.check(self._lib, self._handle, self._lib.sttran_profile_entries(self._handle, arr, n.value, C.byref(n)))
        return [{"kernel": arr[i].kernel.decode(), "class": nat.PROF_NAMES[arr[i].cls], "M": arr[i].M, "N": arr[i].N,
                 "K": arr[i].K, "launches": int(arr[i].launches), "ms": arr[i].ms, "flops": arr[i].flops}
                for i in range(n.value)]

    @property
    def lanes(self):
        return self._lanes

    @lanes.setter
    def lanes(self, n):
        n = int(n)
        if not 1 <= n <= 8:
            raise ValueError("lanes must be 1..8")
        self._lanes = n

    def _sync_lanes(self):
        if self._lanes_set != self._lanes:
            nat.check(self._lib, self._handle, self._lib.sttran_set_lanes(self._handle, self._lanes))
            self._lanes_set = self._lanes          # (sttran_set_lanes synchronised the device: nothing is in flight)
            self._next_lane = 0
            self._drop_inflight(done=True)

    @property
    def pipeline_depth(self):
        """Entries a caller's loop should keep un-joined so that every lane holds a full group: lanes x max(coalesce, 1)."""
        return self._lanes * max(int(self.coalesce), 1)

    def forward_async(self, entry):
        """`forward(entry)` on the next lane (round robin over `model.lanes`), WITHOUT making the current stream wait for
        it: the call returns as soon as the work is enqueued on the lane's own stream (forked from the current stream, so
        whatever produced `entry` there precedes it).  The output tensors are valid for a consumer on the current stream
        only after `model.join(entry)` (or `sync_check()`, which joins every lane).  The loop of tools/test_STTran.py:81-92
        in this form keeps `pipeline_depth` clips in flight:

            pending = collections.deque()
            for entry, gt in loader:
                pending.append((model.forward_async(entry), gt))
                if len(pending) == model.pipeline_depth:
                    pred, g = pending.popleft(); evaluator.evaluate_scene_graph(g, model.join(pred))

        With `model.coalesce = K` (> 1) the call only queues `entry`; the K-th queued entry (or the one that brings the
        queue to `coalesce_max_pairs` pairs) issues the whole group as ONE forward over per-clip pointer tables
        (`pack_clips(group, copy=False)`: nothing is copied) on the next lane, and every entry of the group receives its
        `attention/spatial/contacting_distribution` (+ `distribution`, `pred_labels`, `pred_scores`) as row views of the
        group's outputs -- the same keys, shapes and values as the one-clip call (bit-identical to the packed forward of the
        group).  A `join(entry)` of a still-queued entry issues its (partial) group first: a join never waits for entries
        that have not been submitted.  `sync_check()`, `join()` and a classic `forward` issue what is queued, too.

        `check_indices` is not applied per call here (it would synchronise); index errors raise at `sync_check()`.
        Results are bit-identical to `forward`'s (coalesce off) / to the packed forward of the same group (coalesce on)."""
        if int(self.coalesce) > 1 and not (isinstance(entry, PackedClips) and entry.by_pointer):
            return self._submit(entry)
        self._flush()
        self._async = True
        try:
            return self.forward(entry)
        finally:
            self._async = False

    def _submit(self, entry):
        if self._select:
            from .object_classifier import sgdet_select
            entry = sgdet_select(entry)                      # lib/sttran.py:377 -> :185-283 (per clip: data-dependent sizes)
        P = int(entry["pair_idx"].shape[0])
        if P == 0:
            raise nat.SttranError(3, "entry has no pairs")
        entry["_group"] = None                               # queued: no group yet
        if self._device is not None:
            # whatever produced `entry` on the stream current NOW must precede the group's forward, which is forked from the
            # stream current at flush time (the K-th call, a join, sync_check): remembered per entry, ordered in `_flush`
            entry["_submit_stream"] = torch.cuda.current_stream(torch.device("cuda", self._device))
        self._pending.append(entry)
        self._pending_pairs += P
        if len(self._pending) >= int(self.coalesce) or 0 < int(self.coalesce_max_pairs) <= self._pending_pairs:
            self._flush()
        return entry

    def _flush(self):
        """Issue the queued entries as one by-pointer forward on the next lane and hand every entry its rows."""
        if not self._pending:
            return
        group, self._pending, self._pending_pairs = self._pending, [], 0
        if self._device is not None:
            cur = torch.cuda.current_stream(torch.device("cuda", self._device))
            for st in {e.pop("_submit_stream", cur) for e in group} - {cur}:
                cur.wait_stream(st)                          # entries queued under other streams precede the group forward
        lab = "pred_labels" if self._select else "labels"
        clips = [e if lab == "labels" else dict(e, labels=e[lab]) for e in group]
        packed = pack_clips(clips, copy=False)
        packed.selected = self._select                      # (sgdet without wks: every clip went through sgdet_select)
        self._async = True
        try:
            self.forward(packed)
        except Exception:
            for e in group:
                e.pop("_group", None)
            raise
        finally:
            self._async = False
        p0 = b0 = 0
        dist = packed.get("distribution") if self.mode != "predcls" and not self._select else None
        for e, np_, nb in zip(group, packed["_pairs_per_clip"], packed["_boxes_per_clip"]):
            for k in ("attention_distribution", "spatial_distribution", "contacting_distribution"):
                e[k] = packed[k][p0:p0 + np_]
            for k in ("rel_features", "local_output", "global_output"):
                if "_tap_" + k in packed:
                    e["_tap_" + k] = packed["_tap_" + k][p0:p0 + np_]
            if not self._select:
                e["pred_labels"] = e["labels"]               # lib/sttran.py:91
            if dist is not None:
                e["distribution"] = dist[b0:b0 + nb]         # lib/sttran.py:182-184
                e["pred_scores"] = e["scores"]
            e["_group"], e["_lane"] = packed["_group"], packed["_lane"]
            p0 += np_
            b0 += nb

    def _drop_inflight(self, lane=None, streams=(), done=False):
        """The tensors kept for a lane's last call are released: its group counts as joined -- on `streams` (the raw
        handles that were made to wait for the lane).  `done`: the device was synchronised -- EVERY group issued so far
        (also the ones whose lane has been reused since) is behind every stream: the epoch moves on."""
        for l in (list(self._inflight) if lane is None else [lane]):
            rec = self._inflight.pop(l, None)
            if rec is not None:
                rec[0].joined = True
                rec[0].joined_on.update(streams)
        if done:
            self._epoch += 1

    def join(self, entry=None):
        """Make the current stream wait for the forward that computed `entry` (None: for every lane); returns `entry`.
        A still-queued entry (`coalesce`) is issued first.  Joining an entry whose group was already joined UNDER THIS
        STREAM (another entry of the same coalesced group, or a lane that has been reused since by a call on this stream)
        is free; under another stream it waits for the lane (possibly for a later forward on it: never for less)."""
        if entry is None or entry.get("_group", 0) is None:
            self._flush()
        if self._handle is not None:
            dev = torch.device("cuda", self._device)
            cur = torch.cuda.current_stream(dev).cuda_stream
            if entry is None:
                lanes = [-1] if self._inflight else []
                streams = {rec[0].stream for rec in self._inflight.values()}
            else:
                g = entry.get("_group")
                if g is None or g.epoch < self._epoch or cur in g.joined_on:
                    return entry
                if self._lanes_set <= g.lane:                # (cannot happen: changing the lane count synchronises -> done)
                    return entry
                lanes, streams = [g.lane], ({g.stream} if not g.joined else set())
            for lane in lanes:
                nat.check(self._lib, self._handle, self._lib.sttran_lane_join(self._handle, lane, C.c_void_p(cur)))
                # torch's caching allocator recycles a block on the stream it was ALLOCATED on (the current stream of the
                # forward_async call): when the join happens under another stream, that stream must wait for the lane too
                # before the kept tensors are dropped
                for st in streams - {cur}:
                    nat.check(self._lib, self._handle, self._lib.sttran_lane_join(self._handle, lane, C.c_void_p(st)))
            if entry is None:
                self._drop_inflight(None, streams | {cur})
            else:
                g.joined_on.update(streams | {cur})
                if not g.joined:
                    self._drop_inflight(g.lane, streams | {cur})
        return entry

    _async = False

    def reserve(self, max_pairs, max_boxes):
        self._ensure_handle()
        self._sync_lanes()
        nat.check(self._lib, self._handle, self._lib.sttran_reserve(self._handle, int(max_pairs), int(max_boxes)))

    # ---- forward ---------------------------------------------------------------------------------
    def _dev(self, t, dtype, key="?"):
        dev = torch.device("cuda", self._device)
        ok = isinstance(t, torch.Tensor) and t.device == dev and t.dtype == dtype and t.is_contiguous()
        if ok:
            return t
        if self.strict_inputs:
            what = (f"{tuple(t.shape)} {t.dtype} on {t.device}, contiguous={t.is_contiguous()}"
                    if isinstance(t, torch.Tensor) else type(t).__name__)
            raise ValueError(f"entry[{key!r}] must be a contiguous {dtype} tensor on {dev} (got {what}); "
                             f"strict_inputs=True refuses the hidden copy")
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        nbytes = t.numel() * t.element_size()
        if nbytes > (16 << 20) and key not in self._warned:
            self._warned.add(key)
            warnings.warn(f"STTran: entry[{key!r}] ({nbytes >> 20} MiB) is copied on every call because it is not a "
                          f"contiguous {dtype} tensor on {dev}; fix the producer or set strict_inputs=True to catch it")
        if t.device != dev or t.dtype != dtype:
            t = t.to(device=dev, dtype=dtype)
        return t.contiguous()

    def sync_check(self):
        """Wait for the current stream and raise if a kernel met an out-of-range pair_idx / labels entry since the
        last check (what `check_indices=True` does after every call)."""
        self._flush()
        if self._handle is None:                        # no forward has run yet: nothing enqueued, nothing to check
            return
        stream = torch.cuda.current_stream(torch.device("cuda", self._device)).cuda_stream
        rc = self._lib.sttran_sync_check(self._handle, C.c_void_p(stream))
        self._drop_inflight(done=True)                  # sync_check joined every lane and waited for the device
        if rc == 7:                                     # STTRAN_ERR_INDEX: what torch raises as an IndexError
            msg = self._lib.sttran_last_error(self._handle) or b""
            raise nat.SttranIndexError(rc, msg.decode("utf-8", "replace"))
        nat.check(self._lib, self._handle, rc)

    def forward(self, entry):
        """`STTran.forward` (lib/sttran.py:375-411).  `entry` may carry two optional host-side hints
        that avoid the read-back of `im_idx`: `frame_counts` (pairs per frame) and, for a batch of
        clips packed by `pack_clips`, `clip_num_frames`."""
        self._ensure_handle()
        if self._pending and not self._async:
            self._flush()                                    # queued entries precede a classic forward
        self._sync_lanes()
        lib, h = self._lib, self._handle
        if self._engine_set != self.gemm_engine:
            nat.check(lib, h, lib.sttran_set_gemm_engine(h, {"fp32": 0, "bf16x3": 1, "bf16x3_all": 2}[self.gemm_engine]))
            self._engine_set = self.gemm_engine
        f32, i64 = torch.float32, torch.int64
        if isinstance(entry, PackedClips) and entry.by_pointer:
            return self._forward_by_pointer(entry)
        if self._select:
            from .object_classifier import sgdet_select
            entry = sgdet_select(entry)                      # lib/sttran.py:377 -> :185-283
        feats = self._dev(entry["features"], f32, "features")
        pair = self._dev(entry["pair_idx"], i64, "pair_idx")
        labels = self._dev(entry["pred_labels" if self._select else "labels"], i64, "labels")
        union = self._dev(entry["union_feat"], f32, "union_feat")
        masks = self._dev(entry["spatial_masks"], f32, "spatial_masks")
        P, B = int(pair.shape[0]), int(feats.shape[0])
        if P == 0:
            raise nat.SttranError(3, "entry has no pairs")
        self._check_shapes(feats, pair, labels, union, masks, P, B)
        im = entry["im_idx"]
        im_dtype = nat.DTYPE_I64 if (isinstance(im, torch.Tensor) and not im.dtype.is_floating_point) else nat.DTYPE_F32
        im = self._dev(im, i64 if im_dtype == nat.DTYPE_I64 else f32, "im_idx")
        if tuple(im.shape) != (P,):
            raise ValueError(f"entry['im_idx'] has shape {tuple(im.shape)}, want [{P}]")
        counts = _host_i32(entry.get("frame_counts"))
        clips = _host_i32(entry.get("clip_num_frames"))
        inp = nat.SttranInputs(struct_size=C.sizeof(nat.SttranInputs),
                               num_clips=1 if clips is None else len(clips), num_boxes=B, num_pairs=P,
                               num_frames=0 if counts is None else len(counts), im_idx_dtype=im_dtype)
        if counts is not None:
            inp.frame_counts = counts.ctypes.data_as(C.POINTER(C.c_int32))
        elif "num_frames" in entry:
            inp.num_frames = int(entry["num_frames"])
        if clips is not None:
            inp.clip_num_frames = clips.ctypes.data_as(C.POINTER(C.c_int32))
        inp.features, inp.pair_idx, inp.labels = feats.data_ptr(), pair.data_ptr(), labels.data_ptr()
        inp.union_feat, inp.spatial_masks, inp.im_idx = union.data_ptr(), masks.data_ptr(), im.data_ptr()
        keep = [feats, pair, labels, union, masks, im, counts, clips]
        dist_shape = None
        if self.mode != "predcls" and not self._select:
            boxes = self._dev(entry["boxes"], f32, "boxes")
            dist_in = self._dev(entry["distribution"], f32, "distribution")
            self._check_sgdet(entry, boxes, dist_in, B)
            inp.boxes, inp.distribution = boxes.data_ptr(), dist_in.data_ptr()
            keep += [boxes, dist_in]
            dist_shape = (B, len(self.obj_classes))
        return self._run(entry, inp, P, dist_shape, feats.device, keep)

    def _check_shapes(self, feats, pair, labels, union, masks, P, B, where="entry"):
        # shapes the reference's layers would reject (nn.Linear / Conv2d / .view raise a RuntimeError there)
        if (feats.dim() != 2 or feats.shape[1] != self.feat_dim or tuple(pair.shape) != (P, 2) or tuple(labels.shape) != (B,)
                or tuple(union.shape) != (P, self.feat_dim, 7, 7) or tuple(masks.shape) != (P, 2, 27, 27)):
            raise ValueError(
                f"{where} shapes: features {tuple(feats.shape)} (want [B,{self.feat_dim}]), pair_idx {tuple(pair.shape)} "
                f"(want [P,2]), labels {tuple(labels.shape)} (want [B]), union_feat {tuple(union.shape)} (want "
                f"[P,{self.feat_dim},7,7]), spatial_masks {tuple(masks.shape)} (want [P,2,27,27])")

    def _check_sgdet(self, entry, boxes, dist_in, B):
        # the reference multiplies distribution [B, C-1] with obj_embed.weight [C-1, 200] (lib/sttran.py:174) and
        # would raise on anything else -- e.g. on an entry that already went through forward once, whose
        # `distribution` now holds the [B, C] logits (:182)
        if tuple(dist_in.shape) != (B, len(self.obj_classes) - 1) or tuple(boxes.shape) != (B, 5):
            raise ValueError(f"entry['distribution'] {tuple(dist_in.shape)} / entry['boxes'] {tuple(boxes.shape)}: "
                             f"want [{B},{len(self.obj_classes) - 1}] and [{B},5] (was this entry already forwarded?)")
        if "scores" not in entry or tuple(entry["scores"].shape) != (B,):
            raise ValueError(f"entry['scores'] must be a [{B}] tensor in sgdet mode (lib/sttran.py:184)")

    def _forward_by_pointer(self, packed):
        """A batch of clips handed over as per-clip pointer tables (`pack_clips(entries, copy=False)`): every clip's
        tensors stay where its producer left them (include/sttran_hip.h, SttranInputs form 2)."""
        if self._select and not packed.selected:
            raise NotImplementedError("sgdet without weak supervision selects boxes per clip: forward the clips one by one "
                                      "(or through forward_async with `coalesce`, which selects per clip and batches the rest)")
        lib, f32, i64 = self._lib, torch.float32, torch.int64
        clips = packed.clips
        n = len(clips)
        sg = self.mode != "predcls" and not self._select
        names = ["features", "pair_idx", "labels", "union_feat", "spatial_masks"] + (["boxes", "distribution"] if sg else [])
        tabs = {k: (C.c_void_p * n)() for k in names}
        nb, npairs = (C.c_int64 * n)(), (C.c_int64 * n)()
        keep = [tabs, nb, npairs]
        for i, e in enumerate(clips):
            feats = self._dev(e["features"], f32, "features")
            pair = self._dev(e["pair_idx"], i64, "pair_idx")
            labels = self._dev(e["labels"], i64, "labels")
            union = self._dev(e["union_feat"], f32, "union_feat")
            masks = self._dev(e["spatial_masks"], f32, "spatial_masks")
            Pc, Bc = int(pair.shape[0]), int(feats.shape[0])
            self._check_shapes(feats, pair, labels, union, masks, Pc, Bc, where=f"clip {i}")
            ts = [feats, pair, labels, union, masks]
            if sg:
                boxes = self._dev(e["boxes"], f32, "boxes")
                dist_in = self._dev(e["distribution"], f32, "distribution")
                self._check_sgdet(e, boxes, dist_in, Bc)
                ts += [boxes, dist_in]
            for k, t in zip(names, ts):
                tabs[k][i] = t.data_ptr()
            nb[i], npairs[i] = Bc, Pc
            keep.append(ts)
        P, B = int(sum(packed["_pairs_per_clip"])), int(sum(packed["_boxes_per_clip"]))
        if P == 0:
            raise nat.SttranError(3, "entry has no pairs")
        counts, cl = packed["frame_counts"], packed["clip_num_frames"]
        inp = nat.SttranInputs(struct_size=C.sizeof(nat.SttranInputs), num_clips=n, num_boxes=B, num_pairs=P,
                               num_frames=len(counts), im_idx_dtype=nat.DTYPE_F32)
        inp.frame_counts = counts.ctypes.data_as(C.POINTER(C.c_int32))
        inp.clip_num_frames = cl.ctypes.data_as(C.POINTER(C.c_int32))
        for k in names:
            setattr(inp, "clip_" + k, tabs[k])
        inp.clip_num_boxes, inp.clip_num_pairs = nb, npairs
        keep += [counts, cl]
        return self._run(packed, inp, P, (B, len(self.obj_classes)) if sg else None, torch.device("cuda", self._device), keep)

    def _run(self, entry, inp, P, dist_shape, dev, keep):
        lib, h, f32 = self._lib, self._handle, torch.float32
        att = torch.empty((P, self.attention_class_num), dtype=f32, device=dev)
        spa = torch.empty((P, self.spatial_class_num), dtype=f32, device=dev)
        con = torch.empty((P, self.contact_class_num), dtype=f32, device=dev)
        out = nat.SttranOutputs(struct_size=C.sizeof(nat.SttranOutputs))
        out.attention_distribution, out.spatial_distribution = att.data_ptr(), spa.data_ptr()
        out.contacting_distribution = con.data_ptr()
        dist_out = None
        if dist_shape is not None:
            dist_out = torch.empty(dist_shape, dtype=f32, device=dev)
            out.distribution = dist_out.data_ptr()
        taps = {}
        if self.taps:
            for k in ("rel_features", "local_output", "global_output"):
                taps[k] = torch.empty((P, 1936), dtype=f32, device=dev)
                setattr(out, k + "_tap", taps[k].data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self._async:
            lane = self._next_lane
            self._next_lane = (lane + 1) % self._lanes
            # The lane's stream reads the inputs and writes the outputs after this call has returned.  torch's caching
            # allocator only knows the CURRENT stream: a tensor dropped by the caller would be handed to a later allocation
            # there while the lane is still using it.  So the shim keeps a reference to every tensor of the call until the
            # lane has been joined into the current stream (`join`, `sync_check`) -- or until the lane's next call, which
            # joins it first (that call was issued `lanes` calls ago: the wait is normally already satisfied).
            prev = self._inflight.get(lane)
            if prev is not None:
                nat.check(lib, h, lib.sttran_lane_join(h, lane, C.c_void_p(stream)))
                if prev[0].stream != stream:                 # (see `join`: the allocating stream must have waited as well)
                    nat.check(lib, h, lib.sttran_lane_join(h, lane, C.c_void_p(prev[0].stream)))
                self._drop_inflight(lane, {stream, prev[0].stream})
            nat.check(lib, h, lib.sttran_forward_lane(h, lane, C.byref(inp), C.byref(out), C.c_void_p(stream)))
            group = _Group(lane, stream, self._epoch)
            self._inflight[lane] = (group, keep, att, spa, con, dist_out, taps)
            entry["_lane"], entry["_group"] = lane, group
        else:
            nat.check(lib, h, lib.sttran_forward(h, C.byref(inp), C.byref(out), C.c_void_p(stream)))
        del keep                                             # (the launches read device memory that `entry` keeps alive)
        if self.check_indices and not self._async:
            self.sync_check()
        # ---- the keys the reference writes (lib/sttran.py:91,182-184,404-409) ----
        lazy = isinstance(entry, PackedClips) and entry.by_pointer    # `pred_labels` / `pred_scores` alias on first access
        if not self._select and not lazy:
            entry["pred_labels"] = entry["labels"]
        if dist_out is not None:
            entry["distribution"] = dist_out
            if not lazy:
                entry["pred_scores"] = entry["scores"]
        entry["attention_distribution"] = att
        entry["spatial_distribution"] = spa
        entry["contacting_distribution"] = con
        for k, v in taps.items():
            entry["_tap_" + k] = v
        return entry


class _Group:
    """One forward issued on a lane: which lane, the stream it was forked from (= the stream torch allocated its tensors
    on), whether its kept tensors have been released (`joined`: some consumer stream and the allocating stream are ordered
    behind it) and WHICH streams are ordered behind it (`joined_on`): a `join` of one of its entries is free only under
    one of those (a join under another stream must still wait for the lane)."""
    __slots__ = ("lane", "stream", "joined", "joined_on", "epoch")

    def __init__(self, lane, stream, epoch):
        self.lane, self.stream, self.joined = lane, stream, False
        self.joined_on = set()          # raw stream handles that have been made to wait for this forward
        self.epoch = epoch              # the model's synchronisation epoch it was issued in (see STTran._epoch)


class PackedClips(dict):
    """What `pack_clips` returns: a batch of clips as ONE entry.  With `copy=False` (`by_pointer`) the big tensors are
    NOT concatenated -- `clips` keeps the original per-clip dicts and the model hands their pointers to the library --;
    the small batch-level tensors a consumer of the predictions may ask for (`pair_idx` with batch-global box rows,
    `im_idx` with batch-global frame ids, `labels`, `boxes`, `scores`, `distribution`: a few KB per clip) are
    concatenated on first access (`packed["pair_idx"]`), so a forward that nobody scores never builds them."""

    SMALL = ("labels", "boxes", "scores", "distribution", "pair_idx", "im_idx")

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.clips = None
        self.by_pointer = False
        self.selected = False

    def __missing__(self, key):
        if self.by_pointer and key in ("pred_labels", "pred_scores"):          # lib/sttran.py:91,184: aliases of the inputs
            return self[key[5:]]
        if not self.by_pointer or key not in self.SMALL or not all(key in e for e in self.clips):
            raise KeyError(key)
        if key == "pair_idx":
            offs = np.concatenate(([0], np.cumsum(self["_boxes_per_clip"])[:-1]))
            v = torch.cat([e["pair_idx"] + int(o) for e, o in zip(self.clips, offs)], dim=0)
        elif key == "im_idx":
            offs = np.concatenate(([0], np.cumsum(self["clip_num_frames"])[:-1]))
            v = torch.cat([e["im_idx"] + int(o) for e, o in zip(self.clips, offs)], dim=0)
        else:
            v = torch.cat([e[key] for e in self.clips], dim=0)
        self[key] = v
        return v


def pack_clips(entries, copy=True):
    """Several clips as one `entry` the HIP path processes in a single pass.

    Frames are renumbered consecutively and the host-side `clip_num_frames` / `frame_counts` hints are attached so
    temporal windows stop at clip borders (no reference counterpart: the reference batch is one clip,
    dataloader/wk_action_genome.py:622-627).  `copy=True` concatenates every tensor (box rows of `pair_idx` offset):
    one contiguous entry, at the price of a device-to-device copy of all inputs (4.7 GB for 64 clips of 16x12).
    `copy=False` copies NOTHING: the clips' tensors stay where they are and the library reads them through per-clip
    pointer tables (include/sttran_hip.h, SttranInputs form 2); the entries must stay alive and unmodified until the
    forward has run.  Use `unpack_predictions` to split the outputs again."""
    entries = list(entries)
    cat = PackedClips()
    counts, clips = [], []
    # entries without the `frame_counts` hint (the reference's entries carry none): ONE read-back of their `im_idx`
    # vectors (concatenated on the device: a few hundred bytes per clip), not one synchronisation per clip
    need = [i for i, e in enumerate(entries) if e.get("frame_counts") is None]
    host_im = {}
    if need:
        ims = [entries[i]["im_idx"] for i in need]
        if len(need) > 1 and all(isinstance(t, torch.Tensor) and t.is_cuda for t in ims):
            flat = torch.cat([t.detach().reshape(-1).to(torch.float64) for t in ims]).cpu().numpy().astype(np.int64)
            cuts = np.cumsum([0] + [int(t.numel()) for t in ims])
            host_im = {i: flat[cuts[j]:cuts[j + 1]] for j, i in enumerate(need)}
        else:
            host_im = {i: (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).astype(np.int64)
                       for i, t in zip(need, ims)}
    for i, e in enumerate(entries):
        fc = _host_i32(e.get("frame_counts"))
        if fc is None:
            fr = host_im[i]
            fc = np.bincount(fr, minlength=int(e.get("num_frames", (fr[-1] + 1) if len(fr) else 0))).astype(np.int32)
        counts.append(fc)
        clips.append(len(fc))
    cat["frame_counts"] = np.concatenate(counts) if counts else np.zeros(0, np.int32)
    cat["clip_num_frames"] = np.asarray(clips, dtype=np.int32)
    cat["num_frames"] = int(sum(clips))
    cat["_pairs_per_clip"] = [int(e["pair_idx"].shape[0]) for e in entries]
    cat["_boxes_per_clip"] = [int(e["labels" if "labels" in e else "features"].shape[0]) for e in entries]
    if not copy:
        cat.clips, cat.by_pointer = entries, True
        return cat
    box_off = frame_off = 0
    pair, im = [], []
    for e, nf in zip(entries, clips):
        pair.append(e["pair_idx"] + box_off)
        im.append(e["im_idx"] + frame_off)
        box_off += int(e["labels"].shape[0])
        frame_off += nf
    for k in ("features", "labels", "union_feat", "spatial_masks", "boxes", "scores", "distribution"):
        if all(k in e for e in entries):
            cat[k] = torch.cat([e[k] for e in entries], dim=0)
    cat["pair_idx"] = torch.cat(pair, dim=0)
    cat["im_idx"] = torch.cat(im, dim=0)
    return cat


def unpack_predictions(packed):
    """Split the *_distribution outputs of a packed entry back into one dict per clip."""
    outs = []
    p0 = b0 = 0
    for np_, nb in zip(packed["_pairs_per_clip"], packed["_boxes_per_clip"]):
        d = {k: packed[k][p0:p0 + np_] for k in ("attention_distribution", "spatial_distribution",
                                                 "contacting_distribution")}
        if "distribution" in packed and packed["distribution"].shape[0] == sum(packed["_boxes_per_clip"]):
            d["distribution"] = packed["distribution"][b0:b0 + nb]
        outs.append(d)
        p0 += np_
        b0 += nb
    return outs
