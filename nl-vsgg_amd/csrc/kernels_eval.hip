// Recall@K matching of one clip's predictions against its ground truth, on the device.
//
// What the reference does per frame on the host (`lib/evaluation_recall.py:397-465` builds the frame's
// relation rows, `:209-235` / `:257-300` / `:321-350` pick the candidate triplets of the three
// metrics, `:630-695` orders them by subj_score * obj_score * predicate_score, `:731-773` matches them
// with the ground truth by class triple and two box IoUs, `lib/fpn/box_intersections_cpu/bbox.pyx:21-61`
// is the +1-pixel float64 IoU).  Only the first 50 predictions of a frame can count for R@10/20/50,
// so one workgroup per (frame, metric) finds the 50th-best score by bisection over the order keys in LDS,
// orders the 50 selected candidates, and every thread then checks one ground-truth relation against them.  Integer / byte work, no MFMA; latency-bound.
//
// Ties between exactly equal scores go to the lower candidate index (row-major); the reference leaves
// them to numpy's unstable sort, i.e. to the host CPU it runs on.
#include "kernels.h"

namespace sttran {
namespace {

constexpr int kEvalThreads = 256;
constexpr int kEvalTop = 50;            // R@50 is the deepest list the evaluator reads
constexpr int kEvalMaxCols = 32;
constexpr int kEvalMaxCand = 7488;      // rows * cols of ONE chunk, 58.5 KB of order keys in LDS (26 cols: 288 rows = 96 pairs)
constexpr int kEvalKeysPerThread = (kEvalMaxCand + kEvalThreads - 1) / kEvalThreads;   // 30 order keys in registers

struct EvalArgs {
  const float* att; const float* spa; const float* con;
  const int64_t* pair_idx; const void* im_idx; int im_idx_i64;
  const float* boxes; const int64_t* classes; const float* obj_scores;
  const int32_t* gt_box_off; const float* gt_boxes; const int32_t* gt_classes;
  const int32_t* gt_rel_off; const int32_t* gt_rels;
  uint8_t* flags; int32_t* status;
  int P, B, na, ns, nc;
  double iou_thr;
};

__device__ __forceinline__ double frame_of(const EvalArgs& a, int i) {
  return a.im_idx_i64 ? (double)static_cast<const int64_t*>(a.im_idx)[i] : (double)static_cast<const float*>(a.im_idx)[i];
}

// bbox.pyx:21-61 in float64; no fused multiply-add, so the value equals the host's bit for bit
#pragma clang fp contract(off)
__device__ bool iou_at_least(const float* g, const float* p, double thr) {
  double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
  double iw = fmin(g2, p2) - fmax(g0, p0) + 1.0;
  double ih = fmin(g3, p3) - fmax(g1, p1) + 1.0;
  if (!(iw > 0.0) || !(ih > 0.0)) return 0.0 >= thr;
  double inter = iw * ih;
  double area_g = (g2 - g0 + 1.0) * (g3 - g1 + 1.0);
  double area_p = (p2 - p0 + 1.0) * (p3 - p1 + 1.0);
  double ua = area_g + area_p - inter;
  return inter / ua >= thr;
}

// doubles -> unsigned keys with the same order; 0 is kept for "not a candidate"
__device__ __forceinline__ uint64_t sortable(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint64_t k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  return k ? k : 1;
}

// -DSTTRAN_EVAL_TIMING: workgroup (1,1) leaves the 100 MHz clock at each phase boundary in status[3..8] (tuning aid;
// the caller must then pass at least 9 ints of status)
#ifdef STTRAN_EVAL_TIMING
#define EVAL_STAMP_BEGIN const long long t_start_ = wall_clock64()
#define EVAL_STAMP(k) do { if (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0) a.status[2 + (k)] = (int)(wall_clock64() - t_start_); } while (0)
#else
#define EVAL_STAMP_BEGIN
#define EVAL_STAMP(k)
#endif

struct GtRel { int pred, cs, co; float bs[4], bo[4]; };   // predicate, subject / object class and box

__device__ __forceinline__ GtRel load_gt_rel(const EvalArgs& a, int gb0, int g) {
  GtRel r;
  const int gs = gb0 + a.gt_rels[3 * g], go = gb0 + a.gt_rels[3 * g + 1];
  r.pred = a.gt_rels[3 * g + 2];
  r.cs = a.gt_classes[gs]; r.co = a.gt_classes[go];
#pragma unroll
  for (int d = 0; d < 4; ++d) { r.bs[d] = a.gt_boxes[4 * (size_t)gs + d]; r.bo[d] = a.gt_boxes[4 * (size_t)go + d]; }
  return r;
}

// Workgroup total of per-wave counts (4 waves).  The per-wave count is wave-uniform already (popcount of
// the compare masks, scalar instructions); `part` is double-buffered so one barrier per call suffices.
__device__ __forceinline__ int block_total(int wave_count, int (*part)[4], int& phase) {
  if ((threadIdx.x & 63) == 0) part[phase][threadIdx.x >> 6] = wave_count;
  __syncthreads();
  const int t = part[phase][0] + part[phase][1] + part[phase][2] + part[phase][3];
  phase ^= 1;
  return __builtin_amdgcn_readfirstlane(t);      // the same in every lane: keep it (and what depends on it) scalar
}
__device__ __forceinline__ int wave_count(bool pred) { return __popcll(__ballot(pred)); }

// The (at most) 50 largest keys of key[0..ncand), ties at the 50th value by lowest index: their indices go to
// sel[] in no particular order.  The value of the 50th is found by bisection on the key bits; every thread keeps
// its share of the keys in registers (candidate i lives in thread i % 256, slot i / 256) and counting is a
// compare + popcount of the wave mask + one barrier.  Stops as soon as a threshold splits off exactly 50.
template <int JT>
__device__ void select_top(const uint64_t* key, int ncand, int (*part)[4], int& phase, int* sel, int* nsel) {
  const int tid = threadIdx.x;
  uint64_t kreg[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int i = tid + j * kEvalThreads;
    kreg[j] = i < ncand ? key[i] : 0;
  }
  uint64_t kth = 0;
  bool exact = false;
  for (int bit = 63; bit >= 0; --bit) {
    const uint64_t t = kth | (1ull << bit);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < JT; ++j) cnt += wave_count(kreg[j] >= t);
    const int tot = block_total(cnt, part, phase);
    if (tot >= kEvalTop) {
      kth = t;
      if (tot == kEvalTop) { exact = true; break; }
    }
  }
  int last_tie = ncand;                  // candidates equal to the 50th value count up to this index
  if (kth > 0 && !exact) {
    int cnt = 0;                         // low half: keys above the 50th value, high half: keys equal to it
#pragma unroll
    for (int j = 0; j < JT; ++j) cnt += wave_count(kreg[j] > kth) + (wave_count(kreg[j] == kth) << 16);
    const int tot = block_total(cnt, part, phase);
    const int need = kEvalTop - (tot & 0xffff), ties = tot >> 16;
    if (ties > need) {                   // smallest index I with #{i <= I : key == kth} >= need
      int lo_i = 0, hi_i = ncand - 1;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i) >> 1;
        int c2 = 0;
#pragma unroll
        for (int j = 0; j < JT; ++j) c2 += wave_count((kreg[j] == kth) && (tid + j * kEvalThreads <= mid));
        if (block_total(c2, part, phase) >= need) hi_i = mid; else lo_i = mid + 1;
      }
      last_tie = lo_i;
    }
  }
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int i = tid + j * kEvalThreads;
    const uint64_t k = kreg[j];
    if (k > kth || (kth > 0 && k == kth && i <= last_tie)) sel[atomicAdd(nsel, 1)] = i;
  }
}

__global__ __launch_bounds__(kEvalThreads) void eval_recall_kernel(EvalArgs a) {
  __shared__ uint64_t key[kEvalMaxCand];       // order key of subj_score * obj_score * predicate_score per (row, predicate) of a chunk
  __shared__ int sel[kEvalTop];
  __shared__ uint64_t run_key[2][kEvalTop];     // the running top list, ordered (double-buffered across merges)
  __shared__ int run_idx[2][kEvalTop];          // ... candidate index = row * ncol + predicate over the WHOLE frame
  __shared__ int pr_pred[kEvalTop], pr_cs[kEvalTop], pr_co[kEvalTop];   // the ordered predictions: predicate, classes,
  __shared__ float pr_box[kEvalTop][8];                                  // subject box | object box
  __shared__ int part[2][4], nsel, run_n[2];
  const int f = blockIdx.x, metric = blockIdx.y, tid = threadIdx.x;
  const int ncol = a.na + a.ns + a.nc;
  const int g_lo = a.gt_rel_off[f], g_hi = a.gt_rel_off[f + 1];
  EVAL_STAMP_BEGIN;
  // this thread's first ground-truth relation: loaded now, used in the last phase (the latency hides behind the rest)
  GtRel mine;
  const bool have_mine = g_lo + tid < g_hi;
  if (have_mine) mine = load_gt_rel(a, a.gt_box_off[f], g_lo + tid);
  // the frame's pairs: im_idx is ascending (lib/transformer.py:130-140 assumes the same), so the range is
  // [#{im_idx < f}, #{im_idx < f+1}); counted by the whole workgroup instead of a serial binary search
  int phase = 0;
  int c_lo = 0, c_hi = 0;
  const int p_padded = (a.P + kEvalThreads - 1) / kEvalThreads * kEvalThreads;     // same trip count in every lane
  for (int i = tid; i < p_padded; i += kEvalThreads) {
    const double fr = i < a.P ? frame_of(a, i) : 1e300;
    c_lo += wave_count(fr < (double)f);
    c_hi += wave_count(fr < (double)(f + 1));
  }
  const int lo = block_total(c_lo, part, phase), hi = block_total(c_hi, part, phase);
  const int n = hi - lo;
  const int rows = 3 * n;
  const int chunk_rows = kEvalMaxCand / ncol;                 // 288 rows at 26 predicates
  if (tid == 0) { run_n[0] = 0; run_n[1] = 0; }
  int cur = 0;                                                // which half of run_* holds the list
  // row r of the frame's [3n, ncol] table -> (subject, object) box rows; kind 1 (spatial) swaps them (:429-431)
  auto row_boxes = [&](int r, int& sub, int& obj) {
    const int kind = r / n, p = lo + (r - kind * n);
    const int64_t p0 = a.pair_idx[2 * p], p1 = a.pair_idx[2 * p + 1];
    const bool oob = p0 < 0 || p0 >= a.B || p1 < 0 || p1 >= a.B;
    sub = oob ? 0 : (int)(kind == 1 ? p1 : p0);
    obj = oob ? 0 : (int)(kind == 1 ? p0 : p1);
    return oob;
  };

  for (int r0 = 0; r0 < rows; r0 += chunk_rows) {
    const int crow = min(chunk_rows, rows - r0), ncand = crow * ncol;
    __syncthreads();                                          // the previous chunk's keys / sel are no longer read
    if (tid == 0) nsel = 0;
    EVAL_STAMP(1);
    // ---- the chunk's relation rows: attention | spatial (subject and object swapped) | contacting -------
    for (int rl = tid; rl < crow; rl += kEvalThreads) {
      const int r = r0 + rl;
      const int kind = r / n, p = lo + (r - kind * n);
      int sub, obj;
      const bool oob = row_boxes(r, sub, obj);
      if (oob) atomicOr(a.status, 2);
      // the row of the zero-padded [3n, ncol] score table (lib/evaluation_recall.py:436-441); the softmax of the
      // attention logits (:400) is float32
      float smax = 0.f, ssum = 1.f;
      if (kind == 0) {
        smax = a.att[(size_t)p * a.na];
        for (int c = 1; c < a.na; ++c) smax = fmaxf(smax, a.att[(size_t)p * a.na + c]);
        ssum = 0.f;
        for (int c = 0; c < a.na; ++c) ssum += expf(a.att[(size_t)p * a.na + c] - smax);
      }
      float row[kEvalMaxCols];
#pragma unroll
      for (int c = 0; c < kEvalMaxCols; ++c) {
        float v = 0.f;
        if (kind == 0) { if (c < a.na) v = expf(a.att[(size_t)p * a.na + c] - smax) / ssum; }
        else if (kind == 1) { if (c >= a.na && c < a.na + a.ns) v = a.spa[(size_t)p * a.ns + (c - a.na)]; }
        else { if (c >= a.na + a.ns && c < ncol) v = a.con[(size_t)p * a.nc + (c - a.na - a.ns)]; }
        row[c] = v;
      }
      int arg = 0; float best = row[0];
#pragma unroll
      for (int c = 1; c < kEvalMaxCols; ++c) if (c < ncol && row[c] > best) { best = row[c]; arg = c; }   // first maximum
      uint32_t v;
      if (metric == 0) {                   // with graph constraint: the arg-max predicate of every row (:221-235)
        v = 1u << arg;
      } else if (metric == 1) {            // no constraint: every (row, predicate) entry (:330-340)
        v = ncol == 32 ? 0xffffffffu : ((1u << ncol) - 1u);
      } else {                             // semi constraint (:270-288)
        const bool is_att = ((double)row[0] + (double)row[1]) > 0.0;
        const bool is_multi = !is_att && ((((double)row[3] + (double)row[4]) > 0.0) || (((double)row[9] + (double)row[10]) > 0.0));
        v = 0;
        if (is_att) v = 1u << arg;
        else if (is_multi) {
#pragma unroll
          for (int c = 0; c < kEvalMaxCols; ++c) if (c < ncol && row[c] > 0.5f) v |= 1u << c;
        }
      }
      if (oob) v = 0;
      const double op = oob ? 0.0 : (double)(a.obj_scores[sub] * a.obj_scores[obj]);   // float32 product, then float64 (:663-664)
#pragma unroll
      for (int c = 0; c < kEvalMaxCols; ++c)
        if (c < ncol) key[rl * ncol + c] = ((v >> c) & 1u) ? sortable(op * (double)row[c]) : 0;
    }
    __syncthreads();

    EVAL_STAMP(2);
    // ---- the chunk's 50 best candidates (unordered) -> sel[0..nsel), local indices -----------------------------
    if (ncand <= 4 * kEvalThreads) select_top<4>(key, ncand, part, phase, sel, &nsel);
    else if (ncand <= 8 * kEvalThreads) select_top<8>(key, ncand, part, phase, sel, &nsel);
    else if (ncand <= 16 * kEvalThreads) select_top<16>(key, ncand, part, phase, sel, &nsel);
    else select_top<kEvalKeysPerThread>(key, ncand, part, phase, sel, &nsel);
    __syncthreads();
    EVAL_STAMP(3);
    // ---- merge into the running list: rank of every entry of (running list + this chunk's selection) under
    //      (key descending, frame-wide candidate index ascending); the first 50 survive, already in order ----
    {
      const int nrun = run_n[cur], ns = nsel, tot = nrun + ns;
      const int nxt = cur ^ 1;
      if (tid < tot) {
        const bool from_run = tid < nrun;
        const uint64_t k = from_run ? run_key[cur][tid] : key[sel[tid - nrun]];
        const int i = from_run ? run_idx[cur][tid] : r0 * ncol + sel[tid - nrun];
        int rank = 0;
        for (int j = 0; j < nrun; ++j) {
          const uint64_t k2 = run_key[cur][j];
          rank += (k2 > k) || (k2 == k && run_idx[cur][j] < i);
        }
        for (int j = 0; j < ns; ++j) {
          const uint64_t k2 = key[sel[j]];
          const int i2 = r0 * ncol + sel[j];
          rank += (k2 > k) || (k2 == k && i2 < i);
        }
        if (rank < kEvalTop) { run_key[nxt][rank] = k; run_idx[nxt][rank] = i; }
      }
      if (tid == 0) run_n[nxt] = min(tot, kEvalTop);
      cur = nxt;
    }
  }
  __syncthreads();
  EVAL_STAMP(4);
  const int ntop = run_n[cur];
  if (tid < ntop) {
    const int i = run_idx[cur][tid], r = i / ncol;
    int sub, obj;
    row_boxes(r, sub, obj);
    pr_pred[tid] = i - r * ncol;
    pr_cs[tid] = (int)a.classes[sub]; pr_co[tid] = (int)a.classes[obj];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      pr_box[tid][d] = a.boxes[5 * (size_t)sub + 1 + d];
      pr_box[tid][4 + d] = a.boxes[5 * (size_t)obj + 1 + d];
    }
  }
  __syncthreads();

  EVAL_STAMP(5);
  // ---- one ground-truth relation per thread against the ordered list (:731-773) ------------------------
  const int gb0 = a.gt_box_off[f];
  for (int g = g_lo + tid; g < g_hi; g += kEvalThreads) {
    const GtRel gr = (g == g_lo + tid) ? mine : load_gt_rel(a, gb0, g);
    int first = kEvalTop;
    for (int k = 0; k < ntop; ++k) {
      if (pr_pred[k] != gr.pred || pr_cs[k] != gr.cs || pr_co[k] != gr.co) continue;
      if (!iou_at_least(gr.bs, &pr_box[k][0], a.iou_thr)) continue;
      if (!iou_at_least(gr.bo, &pr_box[k][4], a.iou_thr)) continue;
      first = k;
      break;
    }
    uint8_t* out = a.flags + (size_t)g * 9 + metric * 3;
    out[0] = first < 10; out[1] = first < 20; out[2] = first < 50;
  }
  EVAL_STAMP(6);
}

}  // namespace

hipError_t launch_eval_recall(hipStream_t s, const float* att, const float* spa, const float* con,
                              const int64_t* pair_idx, const void* im_idx, int im_idx_i64, const float* boxes,
                              const int64_t* classes, const float* obj_scores, int P, int B, int na, int ns, int nc,
                              int F, const int32_t* gt_box_off, const float* gt_boxes, const int32_t* gt_classes,
                              const int32_t* gt_rel_off, const int32_t* gt_rels, double iou_thr, uint8_t* flags,
                              int32_t* status) {
  if (F <= 0) return hipSuccess;
  EvalArgs a{att, spa, con, pair_idx, im_idx, im_idx_i64, boxes, classes, obj_scores, gt_box_off, gt_boxes,
             gt_classes, gt_rel_off, gt_rels, flags, status, P, B, na, ns, nc, iou_thr};
  eval_recall_kernel<<<dim3(F, 3), kEvalThreads, 0, s>>>(a);
  return hipGetLastError();
}

// pairs of one frame that fit ONE pass of the key buffer; larger frames are chunked (no limit)
int eval_max_pairs_per_frame(int ncol) { return ncol > 0 ? (kEvalMaxCand / ncol) / 3 : 0; }

}  // namespace sttran
