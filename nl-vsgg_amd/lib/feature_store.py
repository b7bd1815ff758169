"""Reader of the reference's on-disk VinVL feature format + pinned-memory asynchronous upload.

Format (writer: `NL-VSGG/data_preprocess/extract_bbox_features_ag.py:108-120`; reader it replaces:
`lib/assign_pseudo_label.py:27-45::load_feature`): one directory per frame holding
  dets.npy  pickled list of {'class': int, 'conf': float, 'rect': float32[4] (x1,y1,x2,y2)}
  feat.npy  float32 [n_boxes, 2048] box features
`load_feature` returns python lists that are later concatenated and `.cuda()`-copied synchronously.
Here a clip's frames are read straight into ONE pinned staging buffer and sent with a single
non-blocking H2D copy on a side stream, so the copy of clip i+1 overlaps the forward of clip i.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def save_frame_features(frame_dir, classes, confs, rects, feats):
    """Writer in the reference's format (for tests and for producing fixtures)."""
    os.makedirs(frame_dir, exist_ok=True)
    # same statement and field types as extract_bbox_features_ag.py:113-120: a LIST of dicts (numpy builds the object
    # array; an empty list becomes a float64 array of shape (0,)) with numpy-scalar class / conf and a float32[4] rect
    cls_info, conf_info = np.asarray(classes, dtype=np.int64), np.asarray(confs, dtype=np.float32)
    bbox_info = np.asarray(rects, dtype=np.float32).reshape(-1, 4)
    per_box = [{"class": cls_info[i], "conf": conf_info[i], "rect": bbox_info[i]} for i in range(len(cls_info))]
    np.save(os.path.join(frame_dir, "dets.npy"), per_box, allow_pickle=True)
    feat_info = np.asarray(feats, dtype=np.float32)
    if feat_info.ndim != 2:                                     # e.g. an empty python list for a frame without boxes
        feat_info = feat_info.reshape(len(per_box), -1) if len(per_box) else np.zeros((0, 2048), dtype=np.float32)
    np.save(os.path.join(frame_dir, "feat.npy"), feat_info)


def read_frame(frame_dir):
    dets = np.load(os.path.join(frame_dir, "dets.npy"), allow_pickle=True).tolist()
    feat = np.load(os.path.join(frame_dir, "feat.npy"), mmap_mode="r")
    return dets, feat


class ClipFeatureLoader:
    """Loads the per-frame files of a clip into a detector-style record:
         boxes  float32 [B,5] (col 0 = frame index in the clip), classes int64 [B], scores float32 [B],
         features float32 [B, feat_dim] -- `features`/`boxes` on the device when one is given.
    Staging buffers are pinned and reused; two slots let a caller prefetch the next clip."""

    def __init__(self, device=None, feat_dim=2048, slots=2):
        self.device = torch.device(device) if device is not None else None
        self.feat_dim = feat_dim
        self._pin = [None] * slots
        self._ev = [None] * slots
        self._next = 0
        self._stream = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == "cuda" else None

    def _staging(self, rows):
        k = self._next
        self._next = (k + 1) % len(self._pin)
        if self._ev[k] is not None:
            self._ev[k].synchronize()                  # the previous upload from this slot has landed
        need = rows * (self.feat_dim + 5)
        if self._pin[k] is None or self._pin[k].numel() < need:
            buf = torch.empty(max(need, 1), dtype=torch.float32)
            self._pin[k] = buf.pin_memory() if self._stream is not None else buf
        return k, self._pin[k]

    def load(self, frame_dirs):
        frames = [read_frame(d) for d in frame_dirs]
        rows = sum(len(d) for d, _ in frames)
        k, pin = self._staging(rows)
        feats = pin[: rows * self.feat_dim].view(rows, self.feat_dim)
        boxes = pin[rows * self.feat_dim: rows * (self.feat_dim + 5)].view(rows, 5)
        classes = np.empty(rows, dtype=np.int64)
        scores = np.empty(rows, dtype=np.float32)
        o = 0
        fn, bn = feats.numpy(), boxes.numpy()
        for t, (dets, feat) in enumerate(frames):
            n = len(dets)
            if feat.shape != (n, self.feat_dim):
                raise ValueError(f"{frame_dirs[t]}: feat.npy {feat.shape} does not match {n} detections")
            fn[o:o + n] = feat
            for i, dt in enumerate(dets):
                bn[o + i, 0] = t
                bn[o + i, 1:] = dt["rect"]
                classes[o + i] = dt["class"]
                scores[o + i] = dt["conf"]
            o += n
        out = {"classes": torch.from_numpy(classes), "scores": torch.from_numpy(scores), "num_frames": len(frames),
               "boxes_per_frame": [len(d) for d, _ in frames]}
        if self._stream is None:
            out["features"], out["boxes"] = feats.clone(), boxes.clone()
            return out
        consumer = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._stream):
            out["features"] = feats.to(self.device, non_blocking=True)
            out["boxes"] = boxes.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        # The two device tensors were allocated from the side stream's pool.  Tell the caching allocator that the
        # consumer's stream uses them too: otherwise, once the consumer drops them while its (asynchronously enqueued)
        # forward is still running, the next load() could be handed the same block and overwrite it under that forward.
        # A consumer that works on another stream than the one current at load() time must call record_stream itself.
        out["features"].record_stream(consumer)
        out["boxes"].record_stream(consumer)
        self._ev[k] = ev
        out["ready"] = ev        # consumer: torch.cuda.current_stream().wait_event(out['ready'])
        return out
