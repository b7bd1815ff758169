"""Recall@K evaluation of per-clip predictions -- the consumer of the hot path's output dict.

Host-side (numpy) counterpart of the reference's `lib/evaluation_recall.py::SceneGraphEvaluator`
(`:374-465`), written from its scoring contract so that the recall numbers
of both implementations are identical on the same predictions (`tests/test_evaluator.py` checks this
against values captured from the reference, `tests/golden/eval_*.json`).

    ev = SceneGraphEvaluator(mode='predcls', AG_object_classes=..., AG_all_predicates=...,
                             AG_attention_predicates=..., AG_spatial_predicates=...,
                             AG_contacting_predicates=..., iou_threshold=0.5)
    ev.register_container()
    ev.evaluate_scene_graph(gt_annotation, pred)     # once per clip
    ev.calculate_mean_recall(); ev.print_stats(logger)

Five metrics are accumulated in `result_dict` under the reference's keys:
`<mode>_recall` (with graph constraint), `_recall_nogc`, `_semi_recall`, `_mean_recall`,
`_ng_mean_recall` (+ their `_collect` / `_list` helpers).
"""
from __future__ import annotations

import numpy as np

KS = (10, 20, 50)


def _np(x):
    """torch tensor / list / ndarray -> ndarray on the host."""
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def box_iou_plus1(box, boxes):
    """IoU of one box with many, areas measured with the +1-pixel convention and float64
    arithmetic (`lib/fpn/box_intersections_cpu/bbox.pyx:21-61`)."""
    box = np.asarray(box, dtype=np.float64)
    boxes = np.asarray(boxes, dtype=np.float64)
    iw = np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0]) + 1.0
    ih = np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1]) + 1.0
    inter = iw * ih
    area_a = (box[2] - box[0] + 1.0) * (box[3] - box[1] + 1.0)
    area_b = (boxes[:, 2] - boxes[:, 0] + 1.0) * (boxes[:, 3] - boxes[:, 1] + 1.0)
    iou = inter / (area_a + area_b - inter)
    return np.where((iw > 0) & (ih > 0), iou, 0.0)


def match_predictions(gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes, predicate_scores,
                      obj_scores, iou_thresh=0.5, iou_cache=None, ties_by_index=False):
    """For every prediction (in descending order of subj_score * obj_score * predicate_score) the
    list of ground-truth relations it hits: equal class triple and both box IoUs >= iou_thresh
    (`lib/evaluation_recall.py:630-695,731-773`).

    Vectorised: one [n_gt_rel, n_pred_rel] boolean matrix instead of the reference's per-GT Python loop.
    `iou_cache` (optional dict) keeps the [n_gt_box, n_pred_box] IoU table of the frame, which is the
    same for the three metrics computed on it."""
    if pred_rels.size == 0:
        return [[]]
    score = obj_scores[pred_rels[:, 0]] * obj_scores[pred_rels[:, 1]] * predicate_scores
    if ties_by_index:
        order = np.argsort(-score, kind="stable")     # exactly equal scores: lower candidate index first (the device's rule)
    else:
        order = score.argsort()[::-1]                 # same primitive as the reference: ties resolve alike
    pr = pred_rels[order]
    iou = None if iou_cache is None else iou_cache.get("iou")
    if iou is None:
        # the reference rounds the boxes through float32 before the float64 IoU (:757)
        gb = np.asarray(gt_boxes, dtype=np.float32).astype(np.float64)
        pb = np.asarray(pred_boxes, dtype=np.float32).astype(np.float64)
        iou = np.stack([box_iou_plus1(g, pb) for g in gb]) if len(gb) else np.zeros((0, len(pb)))
        if iou_cache is not None:
            iou_cache["iou"] = iou
    same = ((gt_classes[gt_rels[:, 0]][:, None] == pred_classes[pr[:, 0]][None, :])
            & (gt_rels[:, 2][:, None] == pr[:, 2][None, :])
            & (gt_classes[gt_rels[:, 1]][:, None] == pred_classes[pr[:, 1]][None, :]))
    ok = (same & (iou[gt_rels[:, 0]][:, pr[:, 0]] >= iou_thresh) & (iou[gt_rels[:, 1]][:, pr[:, 1]] >= iou_thresh))
    hits = [[] for _ in range(pr.shape[0])]
    gi, pi = np.nonzero(ok)                           # row-major: ascending gt index within a prediction
    for g, i in zip(gi.tolist(), pi.tolist()):
        hits[i].append(g)
    return hits


def _recall_at(hits, n_gt):
    out = {}
    for k in KS:
        got = set()
        for h in hits[:k]:
            got.update(h)
        out[k] = (float(len(got)) / float(n_gt), sorted(got))
    return out


class SceneGraphEvaluator:
    # How predictions with EXACTLY equal scores are ordered.  "numpy" (default) = whatever `ndarray.argsort` does with
    # them, which is what the reference does (lib/evaluation_recall.py:336,670: numpy's unstable sort on the host) -- the
    # outcome is an accident of the sort implementation and of the array length.  "index" = lower candidate index
    # (row-major over the frame's [3 n, predicates] table) first: the documented rule of the device evaluator, which
    # cannot reproduce numpy's accident.  Equal float32 probabilities do occur (one frame in ~1 500 of the AG-shaped test
    # set has such a pair straddling the R@10 cut), so device-vs-host comparisons run the host with "index".
    tie_break = "numpy"

    def __init__(self, mode, AG_object_classes, AG_all_predicates, AG_attention_predicates, AG_spatial_predicates,
                 AG_contacting_predicates, iou_threshold=0.5, constraint=False, semithreshold=None):
        self.mode = mode
        self.result_dict = {}
        self.subject_category = 1
        self.iou_threshold = iou_threshold
        self.AG_object_classes = AG_object_classes
        self.AG_all_predicates = list(AG_all_predicates)
        self.AG_attention_predicates = list(AG_attention_predicates)
        self.AG_spatial_predicates = list(AG_spatial_predicates)
        self.AG_contacting_predicates = list(AG_contacting_predicates)
        self.semithreshold = semithreshold
        self.num_rel = len(self.AG_all_predicates)
        self.hit_flags = None                  # set to [] to record, per frame, which GT relations were hit
        self._att_ix = [self.AG_all_predicates.index(p) for p in self.AG_attention_predicates]
        self._spa_ix = [self.AG_all_predicates.index(p) for p in self.AG_spatial_predicates]
        self._con_ix = [self.AG_all_predicates.index(p) for p in self.AG_contacting_predicates]

    # ---- containers --------------------------------------------------------------------------
    def register_container(self):
        m = self.mode
        for t in ("recall", "recall_nogc", "semi_recall"):
            self.result_dict[f"{m}_{t}"] = {k: [] for k in KS}
        for t in ("mean_recall", "ng_mean_recall"):
            self.result_dict[f"{m}_{t}"] = {k: 0.0 for k in KS}
            self.result_dict[f"{m}_{t}_collect"] = {k: [[] for _ in range(self.num_rel)] for k in KS}
            self.result_dict[f"{m}_{t}_list"] = {k: [] for k in KS}

    # ---- a pack of clips in one call -------------------------------------------------------------
    def evaluate_packed(self, gts, packed_pred):
        """`packed_pred` = the dict the model returned for a `pack_clips` entry, `gts` = the clips' ground truths (lists of
        frames) in pack order.  `pack_clips` numbers the frames of the pack consecutively and offsets `pair_idx`, so the
        packed dict IS one long clip for `evaluate_scene_graph` once the ground-truth lists are chained: same containers,
        same order as one call per clip (`tools/test_STTran.py:88`)."""
        gts = [g if isinstance(g, (list, tuple)) else g.to_annotation(self) for g in gts]
        want = packed_pred.get("clip_num_frames") if hasattr(packed_pred, "get") else None
        if want is not None and [len(g) for g in gts] != [int(x) for x in want]:
            raise ValueError("ground-truth frames per clip do not match the packed entry's clip_num_frames")
        self.evaluate_scene_graph([frame for g in gts for frame in g], packed_pred)

    # ---- per-clip accumulation ------------------------------------------------------------------
    def evaluate_scene_graph(self, gt, pred):
        """`gt`: list over frames of `[ {'person_bbox'}, {'class','bbox','attention_relationship',
        'spatial_relationship','contacting_relationship'}, ... ]` (AG_Test schema,
        `dataloader/wk_action_genome.py:281-292`); `pred`: the dict returned by `STTran.forward`.
        Frame `i` of `gt` is matched with the pairs whose `im_idx == i`.

        One deliberate difference in SIDE EFFECTS: the reference rebinds `pred['attention_distribution']` to its
        soft-max (`lib/evaluation_recall.py:400`), so evaluating the same dict twice soft-maxes twice there.  This
        evaluator leaves `pred` untouched (the same `pred` can go to the host and the device evaluator); a caller that
        read the soft-maxed entry back from the dict applies `softmax(dim=1)` itself."""
        att = _np(pred["attention_distribution"]).astype(np.float32)
        att = att - att.max(axis=1, keepdims=True)      # softmax over the 3 attention logits (:400)
        att = np.exp(att)
        att = att / att.sum(axis=1, keepdims=True)
        spa = _np(pred["spatial_distribution"])
        con = _np(pred["contacting_distribution"])
        pair_idx = _np(pred["pair_idx"])
        im_idx = _np(pred["im_idx"])
        boxes = _np(pred["boxes"])[:, 1:]
        if self.mode == "predcls":
            classes, obj_scores = _np(pred["labels"]), _np(pred["scores"])
        else:
            classes, obj_scores = _np(pred["pred_labels"]), _np(pred["pred_scores"])
        na, ns, nc = att.shape[1], spa.shape[1], con.shape[1]
        for idx, frame_gt in enumerate(gt):
            n_obj = len(frame_gt) - 1
            gt_boxes = np.zeros((n_obj + 1, 4))
            gt_classes = np.zeros(n_obj + 1)
            gt_boxes[0] = np.asarray(_np(frame_gt[0]["person_bbox"])).reshape(-1)[:4]
            gt_classes[0] = self.subject_category
            gt_rels = []
            for m, obj in enumerate(frame_gt[1:], start=1):
                gt_boxes[m] = _np(obj["bbox"])
                gt_classes[m] = obj["class"]
                a = int(np.asarray(_np(obj["attention_relationship"])).reshape(-1)[0])
                gt_rels.append([0, m, self._att_ix[a]])                       # <human, object>
                for sp in np.asarray(_np(obj["spatial_relationship"])).reshape(-1).tolist():
                    gt_rels.append([m, 0, self._spa_ix[int(sp)]])             # <object, human>
                for ct in np.asarray(_np(obj["contacting_relationship"])).reshape(-1).tolist():
                    gt_rels.append([0, m, self._con_ix[int(ct)]])
            gt_rels = np.array(gt_rels)
            sel = im_idx == idx
            pi = pair_idx[sel]
            n = pi.shape[0]
            rels = np.concatenate((pi, pi[:, ::-1], pi), axis=0)             # attention | spatial (reversed) | contact
            scores = np.zeros((3 * n, na + ns + nc))
            scores[:n, :na] = att[sel]
            scores[n:2 * n, na:na + ns] = spa[sel]
            scores[2 * n:, na + ns:] = con[sel]
            self._accumulate(gt_rels, gt_boxes.astype(float), gt_classes, rels, scores,
                             boxes.astype(float), classes, obj_scores)

    def _accumulate(self, gt_rels, gt_boxes, gt_classes, rels, scores, boxes, classes, obj_scores):
        m, n_gt = self.mode, gt_rels.shape[0]
        assert n_gt != 0

        cache = {}

        def run(pred_rels, pscore):
            return match_predictions(gt_rels, gt_boxes, gt_classes, pred_rels, boxes, classes, pscore, obj_scores,
                                     self.iou_threshold, cache, ties_by_index=self.tie_break == "index")

        # with graph constraint: one predicate (the arg-max) per row (:221-235)
        hits_c = run(np.column_stack((rels, scores.argmax(1))), scores.max(1))
        for k, (r, _) in _recall_at(hits_c, n_gt).items():
            self.result_dict[f"{m}_recall"][k].append(r)
        # no graph constraint: the 100 best (row, predicate) entries of obj-score-weighted scores (:351-356)
        overall = (obj_scores[rels].prod(1))[:, None] * scores
        flat = np.argsort(-overall.ravel(), kind="stable" if self.tie_break == "index" else None)[:100]
        ri, ci = np.unravel_index(flat, overall.shape)
        hits_n = run(np.column_stack((rels[ri], ci)), scores[ri, ci])
        for k, (r, _) in _recall_at(hits_n, n_gt).items():
            self.result_dict[f"{m}_recall_nogc"][k].append(r)
        # semi constraint: arg-max for attention rows, every predicate above 0.5 for the others (:270-288)
        is_att = (scores[:, 0] + scores[:, 1]) > 0
        is_multi = ~is_att & (((scores[:, 3] + scores[:, 4]) > 0) | ((scores[:, 9] + scores[:, 10]) > 0))
        over = (scores > 0.5) & is_multi[:, None]
        over[is_att, :] = False
        over[np.nonzero(is_att)[0], scores[is_att].argmax(1)] = True        # arg-max entry of attention rows
        ri_s, ci_s = np.nonzero(over)                                          # row-major = the reference's order
        s_rels = np.column_stack((rels[ri_s], ci_s)) if len(ri_s) else np.zeros((0, 3), dtype=rels.dtype)
        s_sc = scores[ri_s, ci_s]
        hits_s = run(s_rels, s_sc)
        for k, (r, _) in _recall_at(hits_s, n_gt).items():
            self.result_dict[f"{m}_semi_recall"][k].append(r)
        if self.hit_flags is not None:        # [n_gt, 9] table in the device evaluator's layout (tests)
            fl = np.zeros((n_gt, 9), dtype=np.uint8)
            for mi, hits in enumerate((hits_c, hits_n, hits_s)):
                for ki, (_, got) in enumerate(_recall_at(hits, n_gt).values()):
                    fl[got, 3 * mi + ki] = 1
            self.hit_flags.append(fl)
        # per-predicate recall lists for the two mean-recall variants (:126-148); slot 0 also
        # receives the all-predicates count, as in the reference
        for name, hits in (("mean_recall", hits_c), ("ng_mean_recall", hits_n)):
            for k, (_, got) in _recall_at(hits, n_gt).items():
                cnt = np.zeros(self.num_rel); hit = np.zeros(self.num_rel)
                for lab in gt_rels[:, 2]:
                    cnt[int(lab)] += 1; cnt[0] += 1
                for g in got:
                    hit[int(gt_rels[g, 2])] += 1; hit[0] += 1
                col = self.result_dict[f"{m}_{name}_collect"][k]
                for p in range(self.num_rel):
                    if cnt[p] > 0:
                        col[p].append(float(hit[p] / cnt[p]))

    # ---- reductions / report ---------------------------------------------------------------------
    def calculate_mean_recall(self):
        m = self.mode
        for name in ("mean_recall", "ng_mean_recall"):
            for k in KS:
                per = [float(np.mean(v)) if len(v) else 0.0 for v in self.result_dict[f"{m}_{name}_collect"][k]]
                self.result_dict[f"{m}_{name}_list"][k] = per
                self.result_dict[f"{m}_{name}"][k] = sum(per) / float(self.num_rel)

    def summary(self):
        """{metric: {k: value}} with the per-frame lists averaged."""
        m, out = self.mode, {}
        for t in ("recall", "recall_nogc", "semi_recall"):
            out[t] = {k: float(np.mean(v)) if len(v) else float("nan") for k, v in self.result_dict[f"{m}_{t}"].items()}
        for t in ("mean_recall", "ng_mean_recall"):
            out[t] = {k: float(v) for k, v in self.result_dict[f"{m}_{t}"].items()}
        return out

    # ---- merging evaluators that scored disjoint sets of clips (one per rank) ----------------------------------
    # Every reported number is a mean of a per-frame list (Recall@K: over all frames; mean Recall@K: per predicate over
    # the frames that contain it, then over predicates -- `calculate_mean_recall`), so (sum, count) pairs are a sufficient
    # statistic: ranks score their own clips, add the vectors (`lib/distributed.py::all_reduce_recall`: one all-reduce of
    # 2 * (9 + 6 * num_rel) float64) and any rank can print the table of the whole split.  The reference has one process
    # and one evaluator (`tools/test_STTran.py:62-71`); summation order is the only difference (float64, <= 1e-12).
    def partial_sums(self):
        m, rd = self.mode, self.result_dict
        out = []
        for t in ("recall", "recall_nogc", "semi_recall"):
            for k in KS:
                v = rd[f"{m}_{t}"][k]
                out += [float(sum(v)), float(len(v))]        # (the builtin: these are Python lists of floats, 10^4..10^5 long)
        for t in ("mean_recall", "ng_mean_recall"):
            for k in KS:
                for v in rd[f"{m}_{t}_collect"][k]:
                    out += [float(sum(v)), float(len(v))]
        return np.asarray(out, dtype=np.float64)

    def summary_from_partial_sums(self, vec):
        """`summary()` of the union of the evaluators whose `partial_sums()` were added up into `vec`."""
        vec = np.asarray(vec, dtype=np.float64).reshape(-1, 2)
        if vec.shape[0] != 9 + 6 * self.num_rel:
            raise ValueError("partial-sum vector of another evaluator configuration")
        out, i = {}, 0
        for t in ("recall", "recall_nogc", "semi_recall"):
            out[t] = {}
            for k in KS:
                s, n = vec[i]; i += 1
                out[t][k] = float(s / n) if n > 0 else float("nan")
        for t in ("mean_recall", "ng_mean_recall"):
            out[t] = {}
            for k in KS:
                per = [float(s / n) if n > 0 else 0.0 for s, n in vec[i:i + self.num_rel]]
                i += self.num_rel
                out[t][k] = sum(per) / float(self.num_rel)
        return out

    def print_stats(self, logger=None):
        s = self.summary()
        lines = ["======================" + self.mode + "============================"]
        names = {"recall": ("  R", "Recall(Main)"), "recall_nogc": ("  R", "No Graph Constraint Recall(Main)"),
                 "semi_recall": ("  R", "Semi Recall"), "mean_recall": (" mR", "Mean Recall"),
                 "ng_mean_recall": ("ng-mR", "No Graph Constraint Mean Recall")}
        for t, (tag, title) in names.items():
            lines.append("SGG eval: " + "".join(f"{tag} @ {k}: {s[t][k]:.4f}; " for k in KS)
                         + f" for mode={self.mode}, type={title}.")
        text = "\n".join(lines)
        (logger.info if logger is not None else print)(text)
        return text
