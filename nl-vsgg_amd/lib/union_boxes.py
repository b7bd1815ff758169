"""Union boxes and soft box masks of (subject, object) pairs on the device -- the step right before the
hot path (`lib/object_detector.py:110-124`, `lib/draw_rectangles/draw_rectangles.pyx:27-67`), SURVEY 8f-1.

    union_boxes, spatial_masks = union_boxes_and_masks(entry['boxes'], entry['pair_idx'], entry['im_idx'])

replaces the reference's `.cpu().numpy()` -> Cython loop -> `torch.tensor(...).to(device)` round trip;
`spatial_masks` already has the `- 0.5` applied."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native as nat


def union_boxes_and_masks(boxes, pair_idx, im_idx=None, pooling_size=27):
    lib = nat.load()
    if not boxes.is_cuda:
        raise RuntimeError("union_boxes_and_masks runs on the GPU only")
    boxes = boxes.to(torch.float32).contiguous()
    pair_idx = pair_idx.to(device=boxes.device, dtype=torch.int64).contiguous()
    P = int(pair_idx.shape[0])
    im = None if im_idx is None else im_idx.to(device=boxes.device, dtype=torch.float32).contiguous()
    ub = torch.empty((P, 5), dtype=torch.float32, device=boxes.device)
    masks = torch.empty((P, 2, pooling_size, pooling_size), dtype=torch.float32, device=boxes.device)
    stream = torch.cuda.current_stream(boxes.device).cuda_stream
    rc = lib.sttran_union_boxes_masks(C.c_void_p(boxes.data_ptr()), C.c_void_p(pair_idx.data_ptr()),
                                      C.c_void_p(im.data_ptr()) if im is not None else None, P, pooling_size,
                                      C.c_void_p(ub.data_ptr()), C.c_void_p(masks.data_ptr()), C.c_void_p(stream))
    if rc != nat.STTRAN_OK:
        raise nat.SttranError(rc, "sttran_union_boxes_masks failed")
    return ub, masks
