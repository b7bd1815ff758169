"""Portable, seed-addressed synthetic data for the STTran hot path.

Everything here is pure integer / IEEE-754 add-multiply arithmetic (no libm), so the same
(seed, name) produces bit-identical tensors in this container, on the GPU box, and in any
other language that re-implements splitmix64.  Only seeds and small expected outputs are
committed as fixtures; inputs and weights are regenerated from seeds wherever a test runs.

Shapes and distributions follow SURVEY.md §8(d):
  * entry  : the dict `lib/object_detector.py:126-139` (predcls) /
             `lib/assign_pseudo_label.py:1368-1382` (sgdet+wks) hands to `STTran.forward`.
  * weights: the state-dict of `lib/sttran.py:316-372` (+ `lib/transformer.py:104-127`),
             U(+-1/sqrt(fan_in)) with *non-degenerate* LayerNorm / BatchNorm affine parameters
             (default-initialised LayerNorm makes the reference's
             row-sum==0 decoder mask misfire).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

ATTENTION_CLASSES = 3
SPATIAL_CLASSES = 6
CONTACT_CLASSES = 17
NUM_OBJ_CLASSES = 37          # incl. __background__ (dataloader/wk_action_genome.py:214-216)
FEAT_DIM = 2048
EMBED_DIM = 1936
FFN_DIM = 2048
NHEAD = 8
WORD_DIM = 200


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


class Stream:
    """Counter-based splitmix64 stream: value i = mix(key + (i+1)*golden)."""

    def __init__(self, seed: int, name: str):
        k = (int(seed) * 0x9E3779B97F4A7C15 + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF
        self.key = np.uint64(int(_mix(np.array([k], dtype=np.uint64))[0]))
        self.pos = 0

    def _raw(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self.pos + 1, self.pos + n + 1, dtype=np.uint64)
            out = _mix(self.key + idx * _GOLDEN)
        self.pos += n
        return out

    def uniform(self, n: int) -> np.ndarray:
        """float64 in [0,1) with 24 random bits (exactly representable in float32)."""
        return (self._raw(n) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)

    def normal(self, n: int) -> np.ndarray:
        """Irwin-Hall(4) approximation of N(0,1) from the four 16-bit fields of one raw
        draw: integer sum, one float64 multiply -- libm-free, hence bit-portable."""
        chunk = 1 << 22
        out = np.empty(n, dtype=np.float64)
        m16 = np.uint64(0xFFFF)
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            r = self._raw(m)
            tot = ((r & m16) + ((r >> np.uint64(16)) & m16) + ((r >> np.uint64(32)) & m16)
                   + (r >> np.uint64(48))).astype(np.int64)
            out[s:s + m] = (tot - 131070).astype(np.float64) * (1.7320508075688772 / 65536.0)
        return out

    def randint(self, lo: int, hi: int, n: int) -> np.ndarray:
        """integers in [lo, hi] inclusive."""
        span = np.uint64(hi - lo + 1)
        return (self._raw(n) % span).astype(np.int64) + lo


def _u(seed, name, shape, lo, hi):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * Stream(seed, name).uniform(n)).astype(np.float32).reshape(shape)


def _n(seed, name, shape):
    n = int(np.prod(shape))
    return Stream(seed, name).normal(n).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------

def _linear(sd, seed, prefix, out_f, in_f, fan_in=None):
    b = 1.0 / np.sqrt(float(fan_in or in_f))
    sd[prefix + ".weight"] = _u(seed, prefix + ".weight", (out_f, in_f), -b, b)
    sd[prefix + ".bias"] = _u(seed, prefix + ".bias", (out_f,), -b, b)


def _affine(sd, seed, prefix, n):
    sd[prefix + ".weight"] = _u(seed, prefix + ".weight", (n,), 0.5, 1.5)
    sd[prefix + ".bias"] = _u(seed, prefix + ".bias", (n,), -0.5, 0.5)


def _batchnorm(sd, seed, prefix, n):
    _affine(sd, seed, prefix, n)
    sd[prefix + ".running_mean"] = _u(seed, prefix + ".running_mean", (n,), -0.1, 0.1)
    sd[prefix + ".running_var"] = _u(seed, prefix + ".running_var", (n,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = np.array(1, dtype=np.int64)


def _mha(sd, seed, prefix, d):
    b = 1.0 / np.sqrt(float(d))
    sd[prefix + ".in_proj_weight"] = _u(seed, prefix + ".in_proj_weight", (3 * d, d), -b, b)
    sd[prefix + ".in_proj_bias"] = _u(seed, prefix + ".in_proj_bias", (3 * d,), -b, b)
    _linear(sd, seed, prefix + ".out_proj", d, d)


def make_sttran_state_dict(seed: int, enc_layers: int = 1, dec_layers: int = 3,
                           embed_dim: int = EMBED_DIM, ffn_dim: int = FFN_DIM) -> dict:
    """State-dict with the exact keys/shapes of the reference STTran (SURVEY §8b 'Weights')."""
    sd: dict = {}
    # ObjectClassifier (lib/sttran.py:38-51)
    sd["object_classifier.obj_embed.weight"] = _n(seed, "object_classifier.obj_embed.weight",
                                                   (NUM_OBJ_CLASSES - 1, WORD_DIM))
    _batchnorm(sd, seed, "object_classifier.pos_embed.0", 4)
    _linear(sd, seed, "object_classifier.pos_embed.1", 128, 4)
    _linear(sd, seed, "object_classifier.decoder_lin.0", 1024, FEAT_DIM + WORD_DIM + 128)
    _batchnorm(sd, seed, "object_classifier.decoder_lin.1", 1024)
    _linear(sd, seed, "object_classifier.decoder_lin.3", NUM_OBJ_CLASSES, 1024)
    # fusion front-end (lib/sttran.py:336-355)
    b = 1.0 / np.sqrt(float(FEAT_DIM))
    sd["union_func1.weight"] = _u(seed, "union_func1.weight", (256, FEAT_DIM, 1, 1), -b, b)
    sd["union_func1.bias"] = _u(seed, "union_func1.bias", (256,), -b, b)
    b = 1.0 / np.sqrt(2.0 * 49.0)
    sd["conv.0.weight"] = _u(seed, "conv.0.weight", (128, 2, 7, 7), -b, b)
    sd["conv.0.bias"] = _u(seed, "conv.0.bias", (128,), -b, b)
    _batchnorm(sd, seed, "conv.2", 128)
    b = 1.0 / np.sqrt(128.0 * 9.0)
    sd["conv.4.weight"] = _u(seed, "conv.4.weight", (256, 128, 3, 3), -b, b)
    sd["conv.4.bias"] = _u(seed, "conv.4.bias", (256,), -b, b)
    _batchnorm(sd, seed, "conv.6", 256)
    _linear(sd, seed, "subj_fc", 512, FEAT_DIM)
    _linear(sd, seed, "obj_fc", 512, FEAT_DIM)
    _linear(sd, seed, "vr_fc", 512, 256 * 7 * 7)
    sd["obj_embed.weight"] = _n(seed, "obj_embed.weight", (NUM_OBJ_CLASSES, WORD_DIM))
    sd["obj_embed2.weight"] = _n(seed, "obj_embed2.weight", (NUM_OBJ_CLASSES, WORD_DIM))
    # transformer (lib/transformer.py:116-127)
    for i in range(enc_layers):
        p = f"glocal_transformer.local_attention.layers.{i}"
        _mha(sd, seed, p + ".self_attn", embed_dim)
        _linear(sd, seed, p + ".linear1", ffn_dim, embed_dim)
        _linear(sd, seed, p + ".linear2", embed_dim, ffn_dim)
        _affine(sd, seed, p + ".norm1", embed_dim)
        _affine(sd, seed, p + ".norm2", embed_dim)
    for i in range(dec_layers):
        p = f"glocal_transformer.global_attention.layers.{i}"
        _mha(sd, seed, p + ".multihead2", embed_dim)
        _linear(sd, seed, p + ".linear1", ffn_dim, embed_dim)
        _linear(sd, seed, p + ".linear2", embed_dim, ffn_dim)
        _affine(sd, seed, p + ".norm3", embed_dim)
    sd["glocal_transformer.position_embedding.weight"] = _u(
        seed, "glocal_transformer.position_embedding.weight", (2, embed_dim), 0.0, 1.0)
    _linear(sd, seed, "a_rel_compress", ATTENTION_CLASSES, embed_dim)
    _linear(sd, seed, "s_rel_compress", SPATIAL_CLASSES, embed_dim)
    _linear(sd, seed, "c_rel_compress", CONTACT_CLASSES, embed_dim)
    return sd


def sinusoid_table(max_len: int, d_model: int) -> np.ndarray:
    """`PositionalEncoding.pe` (lib/dsg_detr.py:27-36): float32 arithmetic as torch does it."""
    position = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d_model)).astype(np.float32)
    pe = np.zeros((1, max_len, d_model), dtype=np.float32)
    pe[0, :, 0::2] = np.sin(position * div)
    pe[0, :, 1::2] = np.cos(position * div)
    return pe


def _encoder_layer(sd, seed, p, d, ff):
    _mha(sd, seed, p + ".self_attn", d)
    _linear(sd, seed, p + ".linear1", ff, d)
    _linear(sd, seed, p + ".linear2", d, ff)
    _affine(sd, seed, p + ".norm1", d)
    _affine(sd, seed, p + ".norm2", d)


def make_dsg_detr_state_dict(seed: int) -> dict:
    """The tensors of `lib/dsg_detr.py::STTran` (:464-511) that its sgdet forward reads.  The d=2376
    object encoder (`object_classifier.encoder_tran`, 82 M parameters) is never evaluated on that
    branch (`is_wks` is hard-coded, :89,277-288) and is left out; `strict=False` loading ignores it."""
    base = make_sttran_state_dict(seed, enc_layers=0, dec_layers=0)
    sd = {k: v for k, v in base.items() if not k.startswith("glocal_transformer.")}
    sd["positional_encoder.pe"] = sinusoid_table(400, EMBED_DIM)
    _encoder_layer(sd, seed, "local_transformer.layers.0", EMBED_DIM, FFN_DIM)
    for i in range(3):
        _encoder_layer(sd, seed, f"global_transformer.layers.{i}", EMBED_DIM, FFN_DIM)
    return sd


# --------------------------------------------------------------------------------------
# entries
# --------------------------------------------------------------------------------------

def union_box_masks(pair_rois: np.ndarray, size: int = 27) -> np.ndarray:
    """Soft box masks of a (subject, object) box pair inside their union box.

    numpy restatement of `lib/draw_rectangles/draw_rectangles.pyx:27-67` (float32 arithmetic,
    same operation order).  pair_rois [P,8] = (x1,y1,x2,y2) of subject then object.
    Returns [P,2,size,size] in [0,1]; the detector subtracts 0.5 (`lib/object_detector.py:124`).
    """
    r = pair_rois.astype(np.float32)
    f = np.float32
    x1u = np.minimum(r[:, 0], r[:, 4]); y1u = np.minimum(r[:, 1], r[:, 5])
    x2u = np.maximum(r[:, 2], r[:, 6]); y2u = np.maximum(r[:, 3], r[:, 7])
    w = x2u - x1u
    h = y2u - y1u
    grid = np.arange(size, dtype=np.float32)
    out = np.zeros((r.shape[0], 2, size, size), dtype=np.float32)
    clamp = lambda v: np.minimum(np.maximum(v, f(0)), f(1))
    for i in range(2):
        x1b = (r[:, 0 + 4 * i] - x1u) * f(size) / w
        y1b = (r[:, 1 + 4 * i] - y1u) * f(size) / h
        x2b = (r[:, 2 + 4 * i] - x1u) * f(size) / w
        y2b = (r[:, 3 + 4 * i] - y1u) * f(size) / h
        yc = clamp(grid[None, :] + f(1) - y1b[:, None]) * clamp(y2b[:, None] - grid[None, :])
        xc = clamp(grid[None, :] + f(1) - x1b[:, None]) * clamp(x2b[:, None] - grid[None, :])
        out[:, i] = xc[:, None, :] * yc[:, :, None]
    return out


def make_entry(seed: int, pairs_per_frame, boxes_per_frame=None, mode: str = "predcls",
               im_idx_dtype=np.float32, real_masks: bool = False, geometry_only: bool = False) -> dict:
    """One clip in the reference `entry` schema, as numpy arrays.

    pairs_per_frame: list, n_t = number of (person, object) pairs in frame t (may be 0).
    Frame t holds 1 person box followed by n_t object boxes (predcls layout,
    `lib/object_detector.py:57-141`); empty frames hold no boxes.
    geometry_only: leave out `features`, `union_feat`, `spatial_masks` (what the evaluator never reads).
    """
    counts = [int(c) for c in pairs_per_frame]
    T = len(counts)
    boxes, labels, pair_idx, im_idx = [], [], [], []
    lab = Stream(seed, "entry.labels")
    row = 0
    for t, n in enumerate(counts):
        if n == 0:
            continue
        person = row
        labels.append(1)
        row += 1
        for _ in range(n):
            labels.append(int(lab.randint(2, NUM_OBJ_CLASSES - 1, 1)[0]))
            pair_idx.append((person, row))
            im_idx.append(t)
            row += 1
        boxes += [t] * (n + 1)
    B, P = row, len(pair_idx)
    xy = _u(seed, "entry.boxes.xy", (B, 2), 0.0, 300.0)
    wh = _u(seed, "entry.boxes.wh", (B, 2), 10.0, 160.0)
    bx = np.concatenate([np.asarray(boxes, dtype=np.float32)[:, None], xy, xy + wh], axis=1)
    entry = {
        "boxes": bx.astype(np.float32),
        "labels": np.asarray(labels, dtype=np.int64),
        "scores": np.ones(B, dtype=np.float32),
        "pair_idx": np.asarray(pair_idx, dtype=np.int64).reshape(P, 2),
        "im_idx": np.asarray(im_idx, dtype=im_idx_dtype),
        "num_frames": T,
        "frame_counts": np.asarray(counts, dtype=np.int32),
    }
    if not geometry_only:
        entry["features"] = _n(seed, "entry.features", (B, FEAT_DIM))
        entry["union_feat"] = _n(seed, "entry.union_feat", (P, FEAT_DIM, 7, 7))
    if geometry_only:
        pass
    elif real_masks:
        pi = entry["pair_idx"]
        rois = np.concatenate([bx[pi[:, 0], 1:], bx[pi[:, 1], 1:]], axis=1)
        entry["spatial_masks"] = (union_box_masks(rois, 27) - np.float32(0.5)).astype(np.float32)
    else:
        entry["spatial_masks"] = _u(seed, "entry.spatial_masks", (P, 2, 27, 27), -0.5, 0.5)
    if mode != "predcls":
        d = _u(seed, "entry.distribution", (B, NUM_OBJ_CLASSES - 1), 0.0, 1.0)
        d = d / d.sum(axis=1, keepdims=True)
        entry["distribution"] = d.astype(np.float32)
        entry["scores"] = d.max(axis=1).astype(np.float32)
    return entry


def shuffle_boxes(entry: dict, seed: int) -> dict:
    """The same clip with its box rows stored in a random order (per-box arrays permuted, `pair_idx` renumbered;
    the pair order -- and so `im_idx` -- is unchanged).  Box numbers are then no longer monotone in the frame:
    the corner the DSG-DETR position index (`lib/dsg_detr.py:551-555`) treats positionally."""
    B = entry["boxes"].shape[0]
    perm = np.random.RandomState(seed).permutation(B)          # new row r holds old row perm[r]
    inv = np.empty(B, dtype=np.int64)
    inv[perm] = np.arange(B)
    out = dict(entry)
    for k in ("boxes", "labels", "scores", "features", "distribution", "pred_labels", "pred_scores"):
        if k in entry and isinstance(entry[k], np.ndarray) and entry[k].shape[:1] == (B,):
            out[k] = np.ascontiguousarray(entry[k][perm])
    out["pair_idx"] = inv[entry["pair_idx"]]
    return out


def uniform_clip(seed: int, frames: int, boxes: int, **kw) -> dict:
    """T frames x N boxes (1 person + N-1 objects): the BASELINE.json synthetic configs."""
    return make_entry(seed, [boxes - 1] * frames, **kw)


def make_gt_annotation(seed: int, entry: dict) -> list:
    """Synthetic ground truth in the `AG_Test` schema (`dataloader/wk_action_genome.py:281-292`)
    whose boxes coincide with the entry's boxes, so PredCls recall is decided by the
    relation scores alone."""
    st = Stream(seed, "gt.rel")
    gt = []
    bx, pi, fr = entry["boxes"], entry["pair_idx"], entry["im_idx"].astype(np.int64)
    for t in range(int(entry["num_frames"])):
        rows = np.nonzero(fr == t)[0]
        if rows.size == 0:
            continue
        frame = [{"person_bbox": bx[pi[rows[0], 0], 1:][None, :].astype(np.float32)}]
        for p in rows:
            o = pi[p, 1]
            ns = int(st.randint(1, 2, 1)[0]); nc = int(st.randint(1, 2, 1)[0])
            frame.append({
                "class": int(entry["labels"][o]),
                "bbox": bx[o, 1:].astype(np.float32),
                "attention_relationship": st.randint(0, ATTENTION_CLASSES - 1, 1),
                "spatial_relationship": np.unique(st.randint(0, SPATIAL_CLASSES - 1, ns)),
                "contacting_relationship": np.unique(st.randint(0, CONTACT_CLASSES - 1, nc)),
            })
        gt.append(frame)
    return gt


def make_gt_annotation_hard(seed: int, entry: dict, jitter: float = 25.0, flip: float = 0.2) -> list:
    """Ground truth the evaluator has to WORK for (the real SGDet situation): list position == frame id for EVERY
    frame of the clip -- frames without a predicted pair get a person and 1..2 objects of their own, which no
    prediction can hit (`lib/evaluation_recall.py:402` enumerates the list, `:428` selects `im_idx == idx`);
    boxes are moved by up to `jitter` pixels per coordinate, so the 0.5 IoU test decides; a fraction `flip` of the
    object classes differs from the detector's label."""
    st = np.random.RandomState(seed)
    base = make_gt_annotation(seed, entry)
    present = sorted(set(entry["im_idx"].astype(np.int64).tolist()))
    it = iter(base)
    gt = []
    for t in range(int(entry["num_frames"])):
        if t in present:
            frame = next(it)
            frame[0]["person_bbox"] = (frame[0]["person_bbox"] + st.uniform(-jitter, jitter, (1, 4))).astype(np.float32)
            for o in frame[1:]:
                o["bbox"] = (o["bbox"] + st.uniform(-jitter, jitter, 4)).astype(np.float32)
                if st.uniform() < flip:
                    o["class"] = int(st.randint(2, NUM_OBJ_CLASSES))
        else:
            frame = [{"person_bbox": np.array([[10.0, 10.0, 100.0, 200.0]], dtype=np.float32)}]
            for _ in range(int(st.randint(1, 3))):
                x, y = st.uniform(0.0, 200.0, 2)
                frame.append({"class": int(st.randint(2, NUM_OBJ_CLASSES)),
                              "bbox": np.array([x, y, x + 50.0, y + 60.0], dtype=np.float32),
                              "attention_relationship": st.randint(0, ATTENTION_CLASSES, 1),
                              "spatial_relationship": np.unique(st.randint(0, SPATIAL_CLASSES, 2)),
                              "contacting_relationship": np.unique(st.randint(0, CONTACT_CLASSES, 1))})
        gt.append(frame)
    return gt


def make_detector_entry(seed: int, boxes_per_frame, feat_dim: int = 16, fmap_channels: int = 6, fmap_hw=(38, 50),
                        image_wh=(800.0, 600.0)) -> dict:
    """Raw detector output of one clip for the SGDet branch WITHOUT weak supervision (`lib/sttran.py:185-283`,
    SURVEY 8f-2): many overlapping boxes per frame (clusters of jittered copies, so the per-class NMS has work to do),
    a class distribution per box whose arg-max is also the detector's `pred_labels` (classes 5, 8 and 17 occur, so
    `clean_class` duplicates boxes), box features and the backbone feature maps ROIAlign samples.
    boxes_per_frame[t] may be 0 (a frame without detections)."""
    counts = [int(c) for c in boxes_per_frame]
    T, B = len(counts), int(sum(counts))
    frame = np.repeat(np.arange(T), counts).astype(np.float32)
    n_clusters = max(2, B // 3)
    cxy = _u(seed, "det.cluster.xy", (n_clusters, 2), 0.0, 1.0) * np.asarray(image_wh, np.float32) * np.float32(0.7)
    cwh = _u(seed, "det.cluster.wh", (n_clusters, 2), 40.0, 260.0)
    which = Stream(seed, "det.cluster.of").randint(0, n_clusters - 1, B)
    jit = _u(seed, "det.jitter", (B, 4), -14.0, 14.0)
    x1y1 = cxy[which] + jit[:, :2]
    x2y2 = x1y1 + np.maximum(cwh[which] + jit[:, 2:], np.float32(8.0))
    x1y1 = np.maximum(x1y1, np.float32(0.0))
    x2y2 = np.minimum(x2y2, np.asarray(image_wh, np.float32) - np.float32(1.0))
    boxes = np.concatenate([frame[:, None], x1y1, x2y2], axis=1).astype(np.float32)
    d = _u(seed, "det.distribution", (B, NUM_OBJ_CLASSES - 1), 0.0, 1.0)
    # a dominant class per box: the cluster's class (so overlapping boxes compete inside one class), sometimes the
    # person column, sometimes one of the classes clean_class re-labels (5, 8, 17 -> columns 4, 7, 16)
    ccls = Stream(seed, "det.cluster.cls").randint(0, NUM_OBJ_CLASSES - 2, n_clusters)
    special = np.asarray([0, 4, 7, 16, 4, 7])
    pick = Stream(seed, "det.special").randint(0, 9, n_clusters)
    ccls = np.where(pick < special.size, special[np.minimum(pick, special.size - 1)], ccls)
    d[np.arange(B), ccls[which]] += _u(seed, "det.boost", (B,), 1.0, 2.0)
    d = (d / d.sum(axis=1, keepdims=True)).astype(np.float32)
    H, W = fmap_hw
    return {
        "boxes": boxes, "distribution": d, "pred_labels": (np.argmax(d, axis=1) + 1).astype(np.int64),
        "features": _n(seed, "det.features", (B, feat_dim)),
        "fmaps": _n(seed, "det.fmaps", (T, fmap_channels, H, W)),
        "num_frames": T,
    }
