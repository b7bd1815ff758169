"""CPU oracle for SURVEY 8f-2 -- the ObjectClassifier branch of SGDet WITHOUT weak supervision
(`lib/sttran.py:185-283`) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/` may import it.

What it restates, and how each part is pinned:

  * `clean_class` (`lib/sttran.py:52-85`), the per-frame / per-class NMS loop (`:203-237`), label / score /
    human selection (`:239-254`), pair enumeration (`:256-268`), union boxes (`:271-273`): plain Python / torch
    indexing in the reference.  PINNED: `tests/golden/gen_golden_objcls.py` imports the reference class and runs
    exactly that code on seeded inputs (fixtures `tests/golden/objcls_*.npz`), with the two compiled ops below
    stubbed by THIS file's restatements -- so ordering, duplication, tie handling and the empty-frame quirks are
    the reference's own.
  * `nms` (`fasterRCNN/lib/model/csrc/nms.h:10-28` -> `cuda/nms.cu:13-131` on the GPU the reference hard-codes,
    `cpu/nms_cpu.cpp:6-66` on the CPU) and `ROIAlign` forward (`cuda/ROIAlign_cuda.cu:16-118`,
    `cpu/ROIAlign_cpu.cpp:6-217`): compiled extensions.  **PARITY UNPINNED**: the C++ does not compile unmodified
    against this image's PyTorch 2.10 (`AT_DISPATCH_FLOATING_TYPES(dets.type(), ...)` is rejected), the CUDA
    sources cannot be built at all here, and the pure-Python `fasterRCNN/lib/model/nms/nms_cpu.py:20-21` is a
    different (wrong: `np.maximum` for xx2 / yy2) algorithm.  They are restated below from the published source in
    float32 with one rounding per operation (no fused multiply-add), statement by statement.
    The two NMS flavours differ in ONE comparison: `cuda/nms.cu:58` suppresses on IoU `>` threshold,
    `cpu/nms_cpu.cpp:62` on `>=`; `ge=False` (the GPU flavour, what the reference executes) is the default.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


# ---- fasterRCNN/lib/model/csrc/cuda/nms.cu:13-21 (devIoU) / cpu/nms_cpu.cpp:24,53-61 -------------------------------
def _iou_plus1(a, b):
    left, right = max(a[0], b[0]), min(a[2], b[2])
    top, bottom = max(a[1], b[1]), min(a[3], b[3])
    width = max(f32(f32(right - left) + f32(1)), f32(0))
    height = max(f32(f32(bottom - top) + f32(1)), f32(0))
    inter = f32(width * height)
    sa = f32(f32(f32(a[2] - a[0]) + f32(1)) * f32(f32(a[3] - a[1]) + f32(1)))
    sb = f32(f32(f32(b[2] - b[0]) + f32(1)) * f32(f32(b[3] - b[1]) + f32(1)))
    return f32(inter / f32(f32(sa + sb) - inter))


def nms(boxes, scores, threshold, ge=False):
    """Greedy NMS.  boxes [n,4] float32 (x1,y1,x2,y2), scores [n]; returns the kept indices ASCENDING
    (`nms.cu:125-130`: `order_t.index(keep).sort()`; `nms_cpu.cpp:66`: `nonzero(suppressed == 0)`).
    Candidates are visited by descending score; ties by lower index (a stable sort)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    thr = f32(threshold)
    suppressed = np.zeros(n, dtype=bool)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            ovr = _iou_plus1(boxes[i], boxes[j])
            if (ovr >= thr) if ge else (ovr > thr):
                suppressed[j] = True
    return np.nonzero(~suppressed)[0].astype(np.int64)


# ---- fasterRCNN/lib/model/csrc/cuda/ROIAlign_cuda.cu:16-118 (== cpu/ROIAlign_cpu.cpp:16-217) ------------------------
def roi_align(fmaps, rois, pooled=7, spatial_scale=1.0 / 16.0, sampling_ratio=0):
    """fmaps [T,C,H,W] float32, rois [P,5] (batch index, x1,y1,x2,y2) -> [P,C,pooled,pooled].  float32, one rounding
    per operation, the statement order of `RoIAlignForward`."""
    fmaps = np.asarray(fmaps, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    T, C, H, W = fmaps.shape
    P = rois.shape[0]
    out = np.zeros((P, C, pooled, pooled), dtype=np.float32)
    scale = f32(spatial_scale)
    for n in range(P):
        bi = int(rois[n, 0])
        rsw, rsh = f32(rois[n, 1] * scale), f32(rois[n, 2] * scale)
        rew, reh = f32(rois[n, 3] * scale), f32(rois[n, 4] * scale)
        rw, rh = max(f32(rew - rsw), f32(1)), max(f32(reh - rsh), f32(1))
        bh, bw = f32(rh / f32(pooled)), f32(rw / f32(pooled))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f32(rh / f32(pooled))))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f32(rw / f32(pooled))))
        count = f32(gh * gw)
        data = fmaps[bi].reshape(C, H * W)
        for ph in range(pooled):
            for pw in range(pooled):
                acc = np.zeros(C, dtype=np.float32)
                for iy in range(gh):
                    y = f32(f32(rsh + f32(f32(ph) * bh)) + f32(f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh)))
                    for ix in range(gw):
                        x = f32(f32(rsw + f32(f32(pw) * bw)) + f32(f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw)))
                        yy, xx = y, x
                        if yy < f32(-1.0) or yy > f32(H) or xx < f32(-1.0) or xx > f32(W):
                            continue                                   # contributes 0
                        if yy <= 0:
                            yy = f32(0)
                        if xx <= 0:
                            xx = f32(0)
                        y_low, x_low = int(yy), int(xx)
                        if y_low >= H - 1:
                            y_high = y_low = H - 1
                            yy = f32(y_low)
                        else:
                            y_high = y_low + 1
                        if x_low >= W - 1:
                            x_high = x_low = W - 1
                            xx = f32(x_low)
                        else:
                            x_high = x_low + 1
                        ly, lx = f32(yy - f32(y_low)), f32(xx - f32(x_low))
                        hy, hx = f32(f32(1) - ly), f32(f32(1) - lx)
                        w1, w2, w3, w4 = f32(hy * hx), f32(hy * lx), f32(ly * hx), f32(ly * lx)
                        v1, v2 = data[:, y_low * W + x_low], data[:, y_low * W + x_high]
                        v3, v4 = data[:, y_high * W + x_low], data[:, y_high * W + x_high]
                        val = (w1 * v1 + w2 * v2) + w3 * v3 + w4 * v4       # float32 arrays: one rounding per op
                        acc = acc + val
                out[n, :, ph, pw] = acc / count
    return out


# ---- lib/sttran.py:52-85 -----------------------------------------------------------------------------------------
def clean_class(boxes, dist, feats, labels, b, class_idx):
    fb, fd, ff, fl = [], [], [], []
    for i in range(b):
        m = boxes[:, 0] == i
        scores, pred_boxes, f, pl = dist[m], boxes[m], feats[m], labels[m]
        sel = pl == class_idx
        new_scores = scores[sel].copy()
        new_scores[:, class_idx - 1] = 0
        new_labels = (np.argmax(new_scores, axis=1) + 1).astype(np.int64) if new_scores.shape[0] > 0 else np.zeros((0,), np.int64)
        fd += [scores, new_scores]; fb += [pred_boxes, pred_boxes[sel]]; ff += [f, f[sel]]; fl += [pl, new_labels]
    return np.concatenate(fb), np.concatenate(fd), np.concatenate(ff), np.concatenate(fl)


def objcls_select(boxes, dist, feats, pred_labels, nms_threshold=0.6, ge=False, return_sources=False):
    """`ObjectClassifier.forward`, sgdet and `is_wks == False`, up to the pair list (`lib/sttran.py:193-273`).
    boxes [B,5] (col 0 = frame id, ascending), dist [B,36], feats [B,F], pred_labels [B] int64 (the detector's).
    With return_sources, `feats` may be a [B,1] column of row numbers: the gathered column tells which input row each
    output box came from (what the device path returns instead of moving features twice)."""
    boxes = np.asarray(boxes, np.float32); dist = np.asarray(dist, np.float32)
    feats = np.asarray(feats); labels = np.asarray(pred_labels, np.int64)
    b = int(boxes[-1, 0] + 1)
    for c in (5, 8, 17):                                                  # :197-199
        boxes, dist, feats, labels = clean_class(boxes, dist, feats, labels, b, c)
    fb, fd, ff = [], [], []
    for i in range(b):                                                    # :203-237
        m = boxes[:, 0] == i
        scores, pred_boxes, f = dist[m], boxes[m, 1:], feats[m]
        if scores.shape[0] == 0:
            continue
        am = np.argmax(scores, axis=1)
        for j in range(dist.shape[1]):
            inds = np.nonzero(am == j)[0]
            if inds.size == 0:
                continue
            cls_dists, cls_feats = scores[inds], f[inds]
            cls_scores = cls_dists[:, j]
            order = np.argsort(-cls_scores, kind="stable")
            cls_boxes = pred_boxes[inds][order]
            keep = nms(cls_boxes, cls_scores[order], nms_threshold, ge=ge)
            fd.append(cls_dists[order][keep])
            fb.append(np.concatenate([np.full((keep.shape[0], 1), i, np.float32), cls_boxes[keep]], axis=1))
            ff.append(cls_feats[order][keep])
    boxes, dist, feats = np.concatenate(fb), np.concatenate(fd), np.concatenate(ff)
    box_idx = boxes[:, 0].astype(np.int64)
    pred_scores = dist[:, 1:].max(axis=1)                                 # :243-244
    pred_labels = dist[:, 1:].argmax(axis=1).astype(np.int64) + 2
    human = np.zeros((b,), np.int64)                                      # :247-254 (empty frames keep 0)
    gidx = np.arange(boxes.shape[0])
    for i in range(b):
        m = box_idx == i
        if m.any():
            human[i] = gidx[m][np.argmax(dist[m, 0])]
    pred_labels[human] = 1
    pred_scores[human] = dist[human, 0]
    pair, im_idx = [], []                                                 # :256-266
    for j, i in enumerate(human):
        m = box_idx == j
        for k in gidx[m][pred_labels[m] != 1]:
            im_idx.append(j); pair.append([int(i), int(k)])
    pair = np.asarray(pair, np.int64).reshape(-1, 2)
    im_idx = np.asarray(im_idx, np.float32)
    out = {"boxes": boxes, "distribution": dist, "features": feats, "pred_scores": pred_scores.astype(np.float32),
           "pred_labels": pred_labels, "pair_idx": pair, "im_idx": im_idx, "human_idx": human}
    if pair.shape[0]:
        out["union_box"] = np.concatenate([im_idx[:, None], np.minimum(boxes[pair[:, 0], 1:3], boxes[pair[:, 1], 1:3]),
                                           np.maximum(boxes[pair[:, 0], 3:5], boxes[pair[:, 1], 3:5])], axis=1).astype(np.float32)
    return out
