"""SGDet WITHOUT weak supervision: the `else` branch of the reference's
`ObjectClassifier.forward` (`lib/sttran.py:185-283`) on the device.

    entry = sgdet_select(entry)       # what `self.object_classifier(entry)` does when mode == 'sgdet',
                                      # is_wks == False and the module is in eval mode

reads  entry['boxes'] [B,5] (frame id first, sorted by frame), entry['distribution'] [B,36], entry['features'] [B,2048],
       entry['pred_labels'] [B] (the detector's), entry['fmaps'] [T,C,H,W]
writes boxes, distribution, features (clean_class duplicates added, per-class NMS applied), pred_scores, pred_labels,
       pair_idx, im_idx, human_idx, union_box, union_feat (ROIAlign of fmaps), spatial_masks -- the keys and order of
       the reference (`:238-281`).

Three C-ABI calls: `sttran_objcls_select` (one stream synchronisation: the output sizes depend on the data),
`sttran_union_boxes_masks` (f-1) and `sttran_roi_align`.  PyTorch only owns the memory.  No CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native as nat


def roi_align(fmaps, rois, pooled=7, spatial_scale=1.0 / 16.0, sampling_ratio=0):
    """`ROIAlign((7, 7), 1/16, 0)(fmaps, rois)` (`lib/sttran.py:36`): fmaps [T,C,H,W], rois [P,5] -> [P,C,7,7]."""
    lib = nat.load()
    if not (fmaps.is_cuda and fmaps.dtype == torch.float32 and fmaps.is_contiguous() and fmaps.dim() == 4):
        raise ValueError("fmaps must be a contiguous float32 CUDA tensor [T,C,H,W]")
    rois = rois.to(device=fmaps.device, dtype=torch.float32).contiguous()
    T, Cc, H, W = (int(v) for v in fmaps.shape)
    P = int(rois.shape[0])
    out = torch.empty((P, Cc, pooled, pooled), dtype=torch.float32, device=fmaps.device)
    # the entry point has no handle (and so no device of its own): launch under the tensors' device, on its stream
    with torch.cuda.device(fmaps.device):
        stream = torch.cuda.current_stream(fmaps.device).cuda_stream
        rc = lib.sttran_roi_align(C.c_void_p(fmaps.data_ptr()), T, Cc, H, W, C.c_void_p(rois.data_ptr()), P, pooled,
                                  float(spatial_scale), int(sampling_ratio), C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    nat.check(lib, None, rc)
    return out


def sgdet_select(entry, nms_threshold=0.6, nms_ge=False, pooled=7, spatial_scale=1.0 / 16.0, sampling_ratio=0, mask_size=27):
    lib = nat.load()
    boxes = entry["boxes"]
    dev = boxes.device
    if dev.type != "cuda":
        raise RuntimeError("sgdet_select runs on an MI355X only (no CPU path)")
    f32, i64 = torch.float32, torch.int64
    boxes = boxes.to(f32).contiguous()
    dist = entry["distribution"].to(device=dev, dtype=f32).contiguous()
    feats = entry["features"].to(device=dev, dtype=f32).contiguous()
    labels = entry["pred_labels"].to(device=dev, dtype=i64).contiguous()
    B, ncol, F = int(boxes.shape[0]), int(dist.shape[1]), int(feats.shape[1])
    if tuple(boxes.shape) != (B, 5) or dist.shape[0] != B or feats.shape[0] != B or tuple(labels.shape) != (B,) or B == 0:
        raise ValueError("boxes [B,5], distribution [B,C-1], features [B,F] and pred_labels [B] must agree on B > 0")
    T = int(entry["fmaps"].shape[0]) if "fmaps" in entry else int(boxes[:, 0].max().item()) + 1
    return _select(lib, entry, boxes, dist, feats, labels, B, ncol, F, T, nms_threshold, nms_ge, pooled, spatial_scale,
                   sampling_ratio, mask_size, retried=False)


def _select(lib, entry, boxes, dist, feats, labels, B, ncol, F, T, nms_threshold, nms_ge, pooled, spatial_scale, sampling_ratio,
            mask_size, retried):
    dev, f32, i64 = boxes.device, torch.float32, torch.int64
    cap = 4 * B             # a box has at most one copy per clean_class pass and only the newest copy is copied again: 1 + 3
    o = {"boxes": torch.empty((cap, 5), dtype=f32, device=dev), "distribution": torch.empty((cap, ncol), dtype=f32, device=dev),
         "features": torch.empty((cap, F), dtype=f32, device=dev), "pred_scores": torch.empty((cap,), dtype=f32, device=dev),
         "pred_labels": torch.empty((cap,), dtype=i64, device=dev), "pair_idx": torch.empty((cap, 2), dtype=i64, device=dev),
         "im_idx": torch.empty((cap,), dtype=f32, device=dev), "human_idx": torch.zeros((T,), dtype=i64, device=dev)}
    src = torch.empty((cap,), dtype=torch.int32, device=dev)
    nscr = int(lib.sttran_objcls_scratch_bytes(B, T))
    scratch = torch.empty((nscr,), dtype=torch.uint8, device=dev)
    a = nat.SttranObjclsSelect(struct_size=C.sizeof(nat.SttranObjclsSelect), num_frames=T, num_boxes=B, num_cols=ncol, feat_dim=F,
                               nms_threshold=float(nms_threshold), nms_ge=1 if nms_ge else 0, capacity=cap, scratch_bytes=nscr)
    a.boxes, a.distribution, a.features, a.pred_labels = boxes.data_ptr(), dist.data_ptr(), feats.data_ptr(), labels.data_ptr()
    a.out_boxes, a.out_distribution, a.out_features = o["boxes"].data_ptr(), o["distribution"].data_ptr(), o["features"].data_ptr()
    a.out_pred_scores, a.out_pred_labels, a.out_source_row = o["pred_scores"].data_ptr(), o["pred_labels"].data_ptr(), src.data_ptr()
    a.out_pair_idx, a.out_im_idx, a.out_human_idx = o["pair_idx"].data_ptr(), o["im_idx"].data_ptr(), o["human_idx"].data_ptr()
    a.scratch = scratch.data_ptr()
    nb, npair = C.c_int64(0), C.c_int64(0)
    with torch.cuda.device(dev):                                 # handle-less entry point: the tensors' device and stream
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.sttran_objcls_select(C.byref(a), C.byref(nb), C.byref(npair), C.c_void_p(stream))
    if rc == 5 and not retried:
        # STTRAN_ERR_ORDER: rows not grouped by ascending frame id.  The reference picks a frame's rows with
        # `boxes[:, 0] == i` (lib/sttran.py:59-62,205-207), i.e. it takes any order and keeps the order inside a
        # frame: a stable sort by frame id gives it the same rows in the same order
        if float(boxes[:, 0].min().item()) < 0 or not bool((boxes[:, 0] == boxes[:, 0].floor()).all()):
            raise ValueError("entry['boxes'][:, 0] must hold non-negative integer frame ids")
        order = torch.sort(boxes[:, 0], stable=True).indices
        Ts = max(T, int(boxes[:, 0].max().item()) + 1)
        return _select(lib, entry, boxes[order].contiguous(), dist[order].contiguous(), feats[order].contiguous(),
                       labels[order].contiguous(), B, ncol, F, Ts, nms_threshold, nms_ge, pooled, spatial_scale, sampling_ratio,
                       mask_size, retried=True)
    nat.check(lib, None, rc)
    B2, P2 = int(nb.value), int(npair.value)
    # copies of the used rows: the 4 B-row work buffers (features alone: 32 KB per input box) are released on return
    for k in ("boxes", "distribution", "features", "pred_scores", "pred_labels"):
        entry[k] = o[k][:B2].clone()
    entry["pair_idx"], entry["im_idx"] = o["pair_idx"][:P2].clone(), o["im_idx"][:P2].clone()
    entry["human_idx"] = o["human_idx"][:, None]                  # [b, 1] like lib/sttran.py:246
    entry["_source_row"] = src[:B2].clone()
    # union boxes + soft masks (f-1 kernel) and ROIAlign of the backbone feature maps
    from .union_boxes import union_boxes_and_masks
    if P2:
        ub, masks = union_boxes_and_masks(entry["boxes"], entry["pair_idx"], entry["im_idx"], pooling_size=mask_size)
    else:
        ub = torch.empty((0, 5), dtype=f32, device=dev)
        masks = torch.empty((0, 2, mask_size, mask_size), dtype=f32, device=dev)
    entry["union_box"], entry["spatial_masks"] = ub, masks
    if "fmaps" in entry:
        entry["union_feat"] = roi_align(entry["fmaps"], ub, pooled, spatial_scale, sampling_ratio)
    return entry
