"""CPU oracle for the STTran hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference algorithm (rlqja1107/NL-VSGG, `lib/sttran.py` +
`lib/transformer.py` / `lib/transformer_wk.py`), written from the math in SURVEY.md Appendix A.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product path (`nl-vsgg_amd/`) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests or golden vectors, so this oracle is pinned
against outputs of the reference itself: `tests/golden/gen_golden.py` imports
`/root/reference/lib/{sttran,transformer,evaluation_recall}.py` in the build container, runs
them on the seeded inputs of `nl-vsgg_amd/lib/synthetic.py`, and commits the outputs as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against them (<= 2e-5).

Semantics chosen where the reference is ambiguous:
  * key-padding masks are boolean / -inf (`lib/transformer.py:144`), and derived from the
    per-frame pair counts, not from `row-sum == 0` (`lib/transformer.py:161`);
  * frames without pairs are skipped and windows whose two frames are both empty are
    skipped (`lib/transformer_wk.py:144-150,175-185`); with a single frame the encoder output
    is returned (`lib/transformer_wk.py:187-188`).
Because every sequence is processed unpadded, padding never enters the arithmetic.
"""
from __future__ import annotations

import numpy as np

NHEAD = 8
LN_EPS = 1e-5
BN_EPS = 1e-5


def _lin(x, w, b):
    return x @ w.T + b


def _ln(x, g, b):
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (`lib/transformer.py:15-16,46`)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(LN_EPS)) * g + b


def _bn(x, sd, prefix, axis):
    """Eval-mode BatchNorm with running statistics (`lib/sttran.py:340,344`, `:40,49`)."""
    shp = [1] * x.ndim
    shp[axis] = -1
    g = sd[prefix + ".weight"].reshape(shp)
    b = sd[prefix + ".bias"].reshape(shp)
    m = sd[prefix + ".running_mean"].reshape(shp)
    v = sd[prefix + ".running_var"].reshape(shp)
    return (x - m) / np.sqrt(v + x.dtype.type(BN_EPS)) * g + b


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _cast(sd, dtype):
    return {k: (v.astype(dtype, copy=False) if v.dtype.kind == "f" else v) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------
# A7  ObjectClassifier, sgdet + is_wks   (lib/sttran.py:173-184, weights :38-51)
# ---------------------------------------------------------------------------------------
def object_classifier(entry, sd, mode):
    out = {}
    if mode == "predcls":                                   # lib/sttran.py:90-92
        out["pred_labels"] = entry["labels"]
        return out
    dt = sd["subj_fc.weight"].dtype
    dist = entry["distribution"].astype(dt)
    emb = dist @ sd["object_classifier.obj_embed.weight"]   # [B,36] @ [36,200]
    bx = entry["boxes"][:, 1:].astype(dt)
    wh = bx[:, 2:] - bx[:, :2] + dt.type(1.0)               # center_size, lib/fpn/box_utils.py:51-63
    cs = np.concatenate([bx[:, :2] + dt.type(0.5) * wh, wh], axis=1)
    pe = _bn(cs, sd, "object_classifier.pos_embed.0", 1)
    pe = np.maximum(_lin(pe, sd["object_classifier.pos_embed.1.weight"],
                         sd["object_classifier.pos_embed.1.bias"]), 0)
    z = np.concatenate([entry["features"].astype(dt), emb, pe], axis=1)      # [B,2376]
    h = _lin(z, sd["object_classifier.decoder_lin.0.weight"], sd["object_classifier.decoder_lin.0.bias"])
    h = np.maximum(_bn(h, sd, "object_classifier.decoder_lin.1", 1), 0)
    out["distribution"] = _lin(h, sd["object_classifier.decoder_lin.3.weight"],
                               sd["object_classifier.decoder_lin.3.bias"])
    out["pred_labels"] = entry["labels"]
    out["pred_scores"] = entry["scores"]
    return out


# ---------------------------------------------------------------------------------------
# A1  pair fusion   (lib/sttran.py:381-399)
# ---------------------------------------------------------------------------------------
def _im2col(x, k, stride, pad, fill=0.0):
    """x [P,C,H,W] -> [P, C*k*k, Ho*Wo] (channel-major, then ky, kx -- torch weight order)."""
    P, C, H, W = x.shape
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    xp = np.full((P, C, H + 2 * pad, W + 2 * pad), fill, dtype=x.dtype)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    cols = np.empty((P, C, k, k, Ho, Wo), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride]
    return cols.reshape(P, C * k * k, Ho * Wo), Ho, Wo


def _bmm_shared(w, x):
    """w [M,K] @ x [P,K,N] -> [P,M,N] as ONE GEMM [M,K] x [K,P*N] (one BLAS call instead of P small ones)."""
    P, K, N = x.shape
    y = w @ np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(K, P * N)
    return y.reshape(w.shape[0], P, N).transpose(1, 0, 2)


def mask_conv_stack(masks, sd):
    """`self.conv` (lib/sttran.py:337-345): conv7x7/s2/p3 -> ReLU -> BN -> maxpool3/s2/p1
    -> conv3x3/p1 -> ReLU -> BN.  masks [P,2,27,27] -> [P,256,7,7]."""
    P = masks.shape[0]
    cols, Ho, Wo = _im2col(masks, 7, 2, 3)
    w0 = sd["conv.0.weight"].reshape(128, -1)
    c1 = _bmm_shared(w0, cols) + sd["conv.0.bias"][None, :, None]         # [P,128,196]
    c1 = _bn(np.maximum(c1, 0).reshape(P, 128, Ho, Wo), sd, "conv.2", 1)
    pc, Hp, Wp = _im2col(c1, 3, 2, 1, fill=-np.inf)                        # max-pool as im2col
    c2 = pc.reshape(P, 128, 9, Hp * Wp).max(axis=2).reshape(P, 128, Hp, Wp)
    cols2, H2, W2 = _im2col(c2, 3, 1, 1)
    w4 = sd["conv.4.weight"].reshape(256, -1)
    c3 = _bmm_shared(w4, cols2) + sd["conv.4.bias"][None, :, None]
    return _bn(np.maximum(c3, 0).reshape(P, 256, H2, W2), sd, "conv.6", 1)


def pair_fusion(entry, sd, pred_labels, chunk=256):
    dt = sd["subj_fc.weight"].dtype
    feat = entry["features"].astype(dt)
    pi = entry["pair_idx"]
    P = pi.shape[0]
    s = _lin(feat[pi[:, 0]], sd["subj_fc.weight"], sd["subj_fc.bias"])    # :381-382
    o = _lin(feat[pi[:, 1]], sd["obj_fc.weight"], sd["obj_fc.bias"])      # :383-384
    wu = sd["union_func1.weight"].reshape(256, -1)
    vr = np.empty((P, 512), dtype=dt)
    for a in range(0, P, chunk):                                           # :386-387
        b = min(P, a + chunk)
        u = entry["union_feat"][a:b].astype(dt).reshape(b - a, wu.shape[1], 49)
        y = _bmm_shared(wu, u) + sd["union_func1.bias"][None, :, None]     # 1x1 conv
        c3 = mask_conv_stack(entry["spatial_masks"][a:b].astype(dt), sd).reshape(b - a, 256, 49)
        vr[a:b] = _lin((y + c3).reshape(b - a, 256 * 49), sd["vr_fc.weight"], sd["vr_fc.bias"])
    e1 = sd["obj_embed.weight"][pred_labels[pi[:, 0]]]                     # :390-393
    e2 = sd["obj_embed2.weight"][pred_labels[pi[:, 1]]]
    return np.concatenate([s, o, vr, e1, e2], axis=1)                      # :388-399  [P,1936]


# ---------------------------------------------------------------------------------------
# A2  multi-head attention (torch nn.MultiheadAttention semantics).  The three projections and the
#     output projection are row-wise, so they run over ALL tokens at once (as the reference does on
#     its padded batch, lib/transformer.py:144,163); only softmax(QK^T)V is per sequence.
# ---------------------------------------------------------------------------------------
def _attend(q, k, v, seqs, nhead=NHEAD):
    """q,k,v [tokens, d] already projected; seqs = list of (offset, length).  Returns [tokens, d]."""
    d = q.shape[1]
    hd = d // nhead
    scale = q.dtype.type(1.0 / np.sqrt(hd))
    out = np.empty_like(q)
    for o, n in seqs:
        qs = q[o:o + n].reshape(n, nhead, hd).transpose(1, 0, 2) * scale
        ks = k[o:o + n].reshape(n, nhead, hd).transpose(1, 0, 2)
        vs = v[o:o + n].reshape(n, nhead, hd).transpose(1, 0, 2)
        a = np.matmul(qs, ks.transpose(0, 2, 1))
        a = a - a.max(axis=-1, keepdims=True)
        a = np.exp(a)
        a = a / a.sum(axis=-1, keepdims=True)
        out[o:o + n] = np.matmul(a, vs).transpose(1, 0, 2).reshape(n, d)
    return out


def mha(q_in, k_in, v_in, w_in, b_in, w_out, b_out, seqs=None, nhead=NHEAD):
    d = q_in.shape[1]
    q = _lin(q_in, w_in[:d], b_in[:d])
    k = _lin(k_in, w_in[d:2 * d], b_in[d:2 * d])
    v = _lin(v_in, w_in[2 * d:], b_in[2 * d:])
    o = _attend(q, k, v, seqs if seqs is not None else [(0, q.shape[0])], nhead)
    return _lin(o, w_out, b_out)


def encoder_layer(x, sd, p, seqs=None):
    """A3, `TransformerEncoderLayer.forward` lib/transformer.py:20-30 (post-norm, ReLU)."""
    a = mha(x, x, x, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"],
            sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"], seqs)
    h = _ln(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    f = _lin(np.maximum(_lin(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]), 0),
             sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return _ln(h + f, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])


def decoder_layer(g, pos, sd, p, seqs=None):
    """A4, `TransformerDecoderLayer.forward` lib/transformer.py:49-58: q = k = g+pos, v = g;
    LayerNorm after attention only."""
    a = mha(g + pos, g + pos, g, sd[p + ".multihead2.in_proj_weight"], sd[p + ".multihead2.in_proj_bias"],
            sd[p + ".multihead2.out_proj.weight"], sd[p + ".multihead2.out_proj.bias"], seqs)
    h = _ln(g + a, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    f = _lin(np.maximum(_lin(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]), 0),
             sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return h + f


def frame_counts_from_im_idx(im_idx, num_frames=None):
    """`b = int(im_idx[-1] + 1)` (lib/transformer.py:134); rows must be sorted by frame."""
    fr = np.asarray(im_idx).astype(np.int64)
    if fr.size and np.any(np.diff(fr) < 0):
        raise ValueError("im_idx must be sorted ascending (lib/transformer.py:138-140 assumes it)")
    T = int(fr[-1]) + 1 if fr.size else 0
    if num_frames is not None:
        T = max(T, int(num_frames))
    return np.bincount(fr, minlength=T).astype(np.int64)


def transformer(rel, counts, sd, enc_layers, dec_layers, stages=None):
    """A3-A5: spatial encoder per frame, temporal decoder per 2-frame window, 'latter' scatter
    (`lib/transformer.py:130-187`, empty-frame handling `lib/transformer_wk.py:144-195`)."""
    pre = "glocal_transformer."
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    T = len(counts)
    frames = [(int(off[t]), int(counts[t])) for t in range(T) if counts[t] > 0]
    local = rel
    for i in range(enc_layers):                                            # :144
        local = encoder_layer(local, sd, f"{pre}local_attention.layers.{i}", frames)
    if stages is not None:
        stages["local_output"] = local.copy()
    if T < 2:                                                              # transformer_wk.py:187-188
        return local
    pe = sd[pre + "position_embedding.weight"]
    rows, slots, wins = [], [], []                                         # window tokens (:152-159)
    for j in range(T - 1):
        n0, n1 = int(counts[j]), int(counts[j + 1])
        if n0 + n1 == 0:                                                   # transformer_wk.py:175-185
            continue
        wins.append((len(rows), n0 + n1, j, n0))
        rows += list(range(off[j], off[j + 2]))
        slots += [0] * n0 + [1] * n1
    if not wins:
        return local
    g = local[np.asarray(rows)]
    pos = pe[np.asarray(slots)]
    seqs = [(o, n) for o, n, _, _ in wins]
    for i in range(dec_layers):                                            # :163
        g = decoder_layer(g, pos, sd, f"{pre}global_attention.layers.{i}", seqs)
        if stages is not None:
            stages[f"decoder_layer{i}"] = g.copy()
    out = np.zeros_like(rel)
    for o, n, j, n0 in wins:
        if j == 0:                                                         # :181-183
            out[off[0]:off[1]] = g[o:o + n0]
        out[off[j + 1]:off[j + 2]] = g[o + n0:o + n]                       # :185
    return out


def sttran_forward(entry, sd, mode="predcls", enc_layers=1, dec_layers=3, dtype=np.float32,
                   stages=None):
    """`STTran.forward` (lib/sttran.py:375-411).  Returns the keys the reference adds to `entry`."""
    sd = _cast(sd, np.dtype(dtype))
    out = object_classifier(entry, sd, mode)                               # :377
    rel = pair_fusion(entry, sd, out["pred_labels"])                       # :381-399
    if stages is not None:
        stages["rel_features"] = rel.copy()
    counts = frame_counts_from_im_idx(entry["im_idx"], entry.get("num_frames"))
    g = transformer(rel, counts, sd, enc_layers, dec_layers, stages)       # :401
    if stages is not None:
        stages["global_output"] = g.copy()
    out["attention_distribution"] = _lin(g, sd["a_rel_compress.weight"], sd["a_rel_compress.bias"])
    out["spatial_distribution"] = _sigmoid(_lin(g, sd["s_rel_compress.weight"], sd["s_rel_compress.bias"]))
    out["contacting_distribution"] = _sigmoid(_lin(g, sd["c_rel_compress.weight"], sd["c_rel_compress.bias"]))
    return out


# ---------------------------------------------------------------------------------------
# DSG-DETR variant (lib/dsg_detr.py:514-572), sgdet branch (the only one that runs)
# ---------------------------------------------------------------------------------------
def torch_encoder_layer(x, sd, p):
    """stock nn.TransformerEncoderLayer, post-norm, ReLU (lib/dsg_detr.py:21,502-506): the same
    math as `encoder_layer` above under torch's parameter names."""
    return encoder_layer(x, sd, p)


def dsg_detr_sequences(pair_idx, obj_class):
    """Temporal sequences of DSG-DETR (lib/dsg_detr.py:545-555): one per object class present, the
    pairs of that class in pair order.  Position indices are handed out BY POSITION (`:551-554`): the
    sequence's subject boxes are counted (`torch.unique(..., return_counts=True, sorted=True)`) and token i
    gets the index of the count run it falls in -- `[0]*count_0 + [1]*count_1 + ...`.  That equals the dense
    rank of the token's own subject only when the subject numbers ascend along the sequence (boxes stored
    frame by frame); fixture `dsgdetr_shuffled_boxes` holds the other case."""
    seqs, poss = [], []
    for l in np.unique(obj_class):
        k = np.nonzero(obj_class == l)[0]
        _, cnt = np.unique(pair_idx[k, 0], return_counts=True)
        seqs.append(k)
        poss.append(np.repeat(np.arange(len(cnt), dtype=np.int64), cnt))
    return seqs, poss


def dsg_detr_forward(entry, sd, mode="sgdet", dtype=np.float32, stages=None):
    sd = _cast(sd, np.dtype(dtype))
    out = object_classifier(entry, sd, mode)                               # :277-288 (== STTran A7)
    rel = pair_fusion(entry, sd, out["pred_labels"])                       # :517-532
    if stages is not None:
        stages["rel_features"] = rel.copy()
    pi = entry["pair_idx"]
    frames = entry["boxes"][pi[:, 1], 0].astype(np.int64)                  # :536
    if np.any(np.diff(frames) < 0):
        raise ValueError("pairs must be sorted by frame")
    loc = np.empty_like(rel)
    for f in np.unique(frames):                                            # :537-543 spatial encoder
        idx = np.nonzero(frames == f)[0]
        loc[idx] = torch_encoder_layer(rel[idx], sd, "local_transformer.layers.0")
    if stages is not None:
        stages["local_output"] = loc.copy()
    obj_class = out["pred_labels"][pi[:, 1]]
    seqs, poss = dsg_detr_sequences(pi, obj_class)                         # :545-555
    pe = sd["positional_encoder.pe"][0]
    glob = np.zeros_like(rel)
    for k, pos in zip(seqs, poss):                                         # :556-564 temporal encoder
        x = loc[k] + (pe[pos] if mode == "sgdet" else pe[: len(k)])
        for i in range(3):
            x = torch_encoder_layer(x, sd, f"global_transformer.layers.{i}")
        glob[k] = x
    if stages is not None:
        stages["global_output"] = glob.copy()
    out["attention_distribution"] = _lin(glob, sd["a_rel_compress.weight"], sd["a_rel_compress.bias"])
    out["spatial_distribution"] = _sigmoid(_lin(glob, sd["s_rel_compress.weight"], sd["s_rel_compress.bias"]))
    out["contacting_distribution"] = _sigmoid(_lin(glob, sd["c_rel_compress.weight"], sd["c_rel_compress.bias"]))
    return out
