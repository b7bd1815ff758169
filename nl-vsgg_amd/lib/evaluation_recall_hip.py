"""Recall@K evaluation with the per-frame matching on the device.

Same constructor, `register_container / evaluate_scene_graph / calculate_mean_recall / print_stats`
and `result_dict` keys as the reference's `lib/evaluation_recall.py::SceneGraphEvaluator` (`:356-465`)
and as the host evaluator in `evaluation_recall.py`, which is its checker.  `evaluate_scene_graph`
only enqueues one kernel launch per clip (C ABI `sttran_eval_recall`) on the current stream: predictions
stay on the GPU, nothing is copied back and nothing synchronises until the numbers are read
(`result_dict`, `summary()`, `calculate_mean_recall()`, `print_stats()`), when the hit tables of all
pending clips come back in one copy and are tallied with numpy.

Ground truth is static, so it can be packed once and reused across epochs / models:

    packed = ev.pack(gt_annotation)            # or pack_ground_truth(gt_annotation, ev)
    ev.evaluate_scene_graph(packed, pred)

There is no host fallback here: without the HIP library the import of the native module raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _native as nat
from .evaluation_recall import KS, SceneGraphEvaluator, _np


class PackedGroundTruth:
    """One clip's ground truth as flat arrays (frame-major).  Box 0 of every frame is the person."""

    def __init__(self, box_off, boxes, classes, rel_off, rels):
        self.box_off = np.ascontiguousarray(box_off, dtype=np.int32)      # [F+1]
        self.boxes = np.ascontiguousarray(boxes, dtype=np.float32)        # [G,4]; float32 like :757 of the reference
        self.classes = np.ascontiguousarray(classes, dtype=np.int32)      # [G]
        self.rel_off = np.ascontiguousarray(rel_off, dtype=np.int32)      # [F+1]
        self.rels = np.ascontiguousarray(rels, dtype=np.int32).reshape(-1, 3)   # (subject box, object box, predicate)
        self._dev = {}

    @property
    def num_frames(self):
        return len(self.box_off) - 1

    def to_annotation(self, evaluator):
        """Back to the list-of-frames schema `evaluate_scene_graph` of the host evaluator / the reference takes."""
        att, spa, con = evaluator._att_ix, evaluator._spa_ix, evaluator._con_ix
        frames = []
        for f in range(self.num_frames):
            b0, b1 = int(self.box_off[f]), int(self.box_off[f + 1])
            objs = [{"class": int(self.classes[b]), "bbox": self.boxes[b].copy(), "attention_relationship": [],
                     "spatial_relationship": [], "contacting_relationship": []} for b in range(b0 + 1, b1)]
            for s_, o_, p_ in self.rels[int(self.rel_off[f]):int(self.rel_off[f + 1])].tolist():
                if p_ in spa and s_ != 0:
                    objs[s_ - 1]["spatial_relationship"].append(spa.index(p_))
                elif p_ in att:
                    objs[o_ - 1]["attention_relationship"].append(att.index(p_))
                else:
                    objs[o_ - 1]["contacting_relationship"].append(con.index(p_))
            frames.append([{"person_bbox": self.boxes[b0][None, :].copy()}] + objs)
        return frames

    @staticmethod
    def concat(parts):
        """Ground truth of several clips as ONE (frames in order): the counterpart of `pack_clips` on the prediction
        side, so a packed forward can go to the evaluator in one call (frame f of clip c becomes frame
        sum(frames of clips < c) + f, exactly how `pack_clips` renumbers `im_idx`)."""
        parts = list(parts)
        box_off, rel_off = [np.zeros(1, np.int32)], [np.zeros(1, np.int32)]
        b = r = 0
        for p in parts:
            box_off.append(p.box_off[1:] + b)
            rel_off.append(p.rel_off[1:] + r)
            b += int(p.box_off[-1]); r += int(p.rel_off[-1])
        return PackedGroundTruth(np.concatenate(box_off), np.concatenate([p.boxes for p in parts]).reshape(-1, 4),
                                 np.concatenate([p.classes for p in parts]), np.concatenate(rel_off),
                                 np.concatenate([p.rels for p in parts]).reshape(-1, 3))

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            # one upload: [box_off | rel_off | classes | rels] as int32, boxes as float32
            ints = np.concatenate((self.box_off, self.rel_off, self.classes, self.rels.reshape(-1)))
            ti = torch.from_numpy(ints).to(device)
            tb = torch.from_numpy(self.boxes).to(device)
            F, G = self.num_frames, len(self.classes)
            self._dev[key] = (ti[:F + 1], ti[F + 1:2 * F + 2], ti[2 * F + 2:2 * F + 2 + G], ti[2 * F + 2 + G:], tb)
        return self._dev[key]


def pack_ground_truth(gt, evaluator):
    """`gt`: list over frames of `[{'person_bbox'}, {'class','bbox','attention_relationship',
    'spatial_relationship','contacting_relationship'}, ...]` (`dataloader/wk_action_genome.py:281-292`);
    relation rows in the order of `lib/evaluation_recall.py:417-431`."""
    att_ix, spa_ix, con_ix = evaluator._att_ix, evaluator._spa_ix, evaluator._con_ix
    box_off, rel_off, boxes, classes, rels = [0], [0], [], [], []
    for frame_gt in gt:
        boxes.append(np.asarray(_np(frame_gt[0]["person_bbox"]), dtype=np.float64).reshape(-1)[:4])
        classes.append(evaluator.subject_category)
        for m, obj in enumerate(frame_gt[1:], start=1):
            boxes.append(np.asarray(_np(obj["bbox"]), dtype=np.float64).reshape(-1)[:4])
            classes.append(int(obj["class"]))
            a = int(np.asarray(_np(obj["attention_relationship"])).reshape(-1)[0])
            rels.append((0, m, att_ix[a]))
            for sp in np.asarray(_np(obj["spatial_relationship"])).reshape(-1).tolist():
                rels.append((m, 0, spa_ix[int(sp)]))
            for ct in np.asarray(_np(obj["contacting_relationship"])).reshape(-1).tolist():
                rels.append((0, m, con_ix[int(ct)]))
        box_off.append(len(classes))
        rel_off.append(len(rels))
    return PackedGroundTruth(box_off, np.asarray(boxes, dtype=np.float64).reshape(-1, 4), classes, rel_off,
                             np.asarray(rels, dtype=np.int64).reshape(-1, 3))


def tally_hit_flags(evaluator, result_dict, packed, flags):
    """Turn the [num_gt_rels, 9] hit table of one clip, or of a list of clips (tables concatenated in the
    same order), into the reference's containers (`lib/evaluation_recall.py:221-235` recall lists,
    `:69-93` / `:146-170` mean-recall collections).  Frames are appended in order, so the containers
    equal what per-clip, per-frame calls would have produced."""
    clips = packed if isinstance(packed, (list, tuple)) else [packed]
    m, nrel = evaluator.mode, evaluator.num_rel
    n_gt = np.concatenate([np.diff(c.rel_off) for c in clips]).astype(np.int64)
    labels = np.concatenate([c.rels[:, 2] for c in clips]).astype(np.int64)
    if (n_gt <= 0).any():
        raise AssertionError("a frame without ground-truth relations (the reference asserts the same)")
    starts = np.concatenate(([0], np.cumsum(n_gt)[:-1]))
    fl = flags.astype(np.int64)
    sums = np.add.reduceat(fl, starts, axis=0)                                           # [F, 9]
    ratio = sums / n_gt[:, None]
    for mi, t in enumerate(("recall", "recall_nogc", "semi_recall")):
        for ki, k in enumerate(KS):
            result_dict[f"{m}_{t}"][k].extend(ratio[:, 3 * mi + ki].tolist())
    F = len(n_gt)
    key = np.repeat(np.arange(F, dtype=np.int64), n_gt) * nrel + labels
    cnt = np.bincount(key, minlength=F * nrel).reshape(F, nrel).astype(np.float64)
    cnt[:, 0] += n_gt                                    # slot 0 also receives every relation, as in the reference
    present = [np.nonzero(cnt[:, p] > 0)[0] for p in range(nrel)]
    for mi, name in ((0, "mean_recall"), (1, "ng_mean_recall")):
        for ki, k in enumerate(KS):
            hit = np.bincount(key, weights=fl[:, 3 * mi + ki], minlength=F * nrel).reshape(F, nrel)
            hit[:, 0] += sums[:, 3 * mi + ki]
            col = result_dict[f"{m}_{name}_collect"][k]
            for p in range(nrel):
                if present[p].size:
                    col[p].extend((hit[present[p], p] / cnt[present[p], p]).tolist())


class SceneGraphEvaluator_HIP(SceneGraphEvaluator):
    """Drop-in for `SceneGraphEvaluator` whose per-frame matching runs on the GPU."""

    def __init__(self, *args, **kwargs):
        self._rd = {}
        self._pending = []
        self._status = None
        super().__init__(*args, **kwargs)
        self._lib = nat.load()
        self.max_pairs_per_frame = int(self._lib.sttran_eval_max_pairs(self.num_rel))

    # `result_dict` reads as the reference's attribute; reading it brings pending clips in first
    @property
    def result_dict(self):
        self.flush()
        return self._rd

    @result_dict.setter
    def result_dict(self, value):
        self._rd = value

    def pack(self, gt):
        return pack_ground_truth(gt, self)

    def evaluate_scene_graph(self, gt, pred):
        self._poll()
        packed = gt if isinstance(gt, PackedGroundTruth) else pack_ground_truth(gt, self)
        att = pred["attention_distribution"]
        if not (hasattr(att, "is_cuda") and att.is_cuda):
            raise RuntimeError("SceneGraphEvaluator_HIP needs the predictions on the GPU (no CPU path here)")
        dev = att.device

        def on_dev(x, dtype):
            t = x if hasattr(x, "is_cuda") else torch.from_numpy(np.ascontiguousarray(_np(x)))
            return t.to(device=dev, dtype=dtype).contiguous()

        att = on_dev(att, torch.float32)
        spa = on_dev(pred["spatial_distribution"], torch.float32)
        con = on_dev(pred["contacting_distribution"], torch.float32)
        pair = on_dev(pred["pair_idx"], torch.int64)
        im = pred["im_idx"]
        im_i64 = (im.dtype in (torch.int64, torch.int32)) if hasattr(im, "is_cuda") else np.issubdtype(_np(im).dtype, np.integer)
        im = on_dev(im, torch.int64 if im_i64 else torch.float32)
        boxes = on_dev(pred["boxes"], torch.float32)
        if self.mode == "predcls":
            classes, scores = on_dev(pred["labels"], torch.int64), on_dev(pred["scores"], torch.float32)
        else:
            classes, scores = on_dev(pred["pred_labels"], torch.int64), on_dev(pred["pred_scores"], torch.float32)
        P, B = int(pair.shape[0]), int(boxes.shape[0])
        if att.shape != (P, len(self._att_ix)) or spa.shape != (P, len(self._spa_ix)) or con.shape != (P, len(self._con_ix)):
            raise ValueError("prediction tables do not match pair_idx / the predicate lists")
        if classes.shape[0] != B or scores.shape[0] != B or im.shape[0] != P:
            raise ValueError("boxes / labels / scores / im_idx lengths disagree")
        box_off, rel_off, gcls, grel, gbox = packed.on(dev)
        R = int(packed.rels.shape[0])
        flags = torch.empty((R, 9), dtype=torch.uint8, device=dev)
        if self._status is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        inp = nat.SttranEvalInputs(
            C.sizeof(nat.SttranEvalInputs), packed.num_frames, P, B, R,
            len(self._att_ix), len(self._spa_ix), len(self._con_ix),
            nat.DTYPE_I64 if im_i64 else nat.DTYPE_F32, 0, float(self.iou_threshold),
            att.data_ptr(), spa.data_ptr(), con.data_ptr(), pair.data_ptr(), im.data_ptr(), boxes.data_ptr(),
            classes.data_ptr(), scores.data_ptr(), box_off.data_ptr(), gbox.data_ptr(), gcls.data_ptr(),
            rel_off.data_ptr(), grel.data_ptr())
        rc = self._lib.sttran_eval_recall(C.byref(inp), C.c_void_p(flags.data_ptr()), C.c_void_p(self._status.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != nat.STTRAN_OK:
            raise nat.SttranError(rc, "sttran_eval_recall failed")
        # the launch reads these tensors asynchronously: keep them alive until the flush
        host = done = None
        if R * 9 >= self.EAGER_BYTES:
            # a big table (a packed call): start its copy-back now and tally it as soon as it has landed -- under the
            # forwards that follow -- instead of leaving all host work to the end of the loop
            host = torch.empty((R, 9), dtype=torch.uint8, pin_memory=True)
            host.copy_(flags, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        self._pending.append((packed, flags, (att, spa, con, pair, im, boxes, classes, scores), host, done))

    EAGER_BYTES = 1 << 16

    def _poll(self):
        """Tally, in order, the pending tables whose copy-back has completed (never blocks)."""
        while self._pending and self._pending[0][4] is not None and self._pending[0][4].query():
            packed, _, _, host, _ = self._pending.pop(0)
            tally_hit_flags(self, self._rd, packed, host.numpy())

    def evaluate_packed(self, gts, packed_pred):
        """A `pack_clips` forward scored in ONE call: `gts` = the clips' ground truths in pack order (lists of frames or
        PackedGroundTruth; or one PackedGroundTruth already made by `PackedGroundTruth.concat`), `packed_pred` = the dict
        the model returned for the packed entry.  Same containers, same order of the per-frame lists as one
        `evaluate_scene_graph` call per clip (the reference's loop, `tools/test_STTran.py:88`) -- at 1/64 of the host
        work per clip."""
        if isinstance(gts, PackedGroundTruth):
            packed = gts
        else:
            parts = [g if isinstance(g, PackedGroundTruth) else pack_ground_truth(g, self) for g in gts]
            want = packed_pred.get("clip_num_frames")
            if want is not None and [p.num_frames for p in parts] != [int(x) for x in want]:
                raise ValueError("ground-truth frames per clip do not match the packed entry's clip_num_frames")
            packed = PackedGroundTruth.concat(parts)
        if "num_frames" in packed_pred and packed.num_frames != int(packed_pred["num_frames"]):
            raise ValueError("ground truth and packed prediction disagree on the number of frames")
        self.evaluate_scene_graph(packed, packed_pred)

    def flush(self):
        """Copy the pending hit tables back (one D2H) and add them to the containers."""
        if not self._pending:
            return
        pending, self._pending = self._pending, []
        status = int(self._status.item())                                  # synchronises
        self._status.zero_()
        if status & 2:
            raise IndexError("pair_idx out of range of boxes")
        lazy = []                                  # consecutive small tables: one concatenated copy-back

        def drain():
            if lazy:
                tally_hit_flags(self, self._rd, [p[0] for p in lazy], torch.cat([p[1] for p in lazy]).cpu().numpy())
                lazy.clear()
        for p in pending:                          # in call order: the per-frame lists must keep it
            if p[3] is None:
                lazy.append(p)
            else:
                drain()
                p[4].synchronize()
                tally_hit_flags(self, self._rd, p[0], p[3].numpy())
        drain()
