// kernels_objcls.hip -- SURVEY 8f-2: the ObjectClassifier branch of SGDet WITHOUT weak supervision
// (lib/sttran.py:185-283): clean_class (:52-85), per-(frame, class) greedy NMS at 0.6 (:203-237 with
// fasterRCNN/lib/model/csrc/cuda/nms.cu:13-131), label / score / human selection (:239-254), pair enumeration
// (:256-268) and ROIAlign of the union boxes on the detector's feature maps (:275; the op is
// fasterRCNN/lib/model/csrc/cuda/ROIAlign_cuda.cu:65-118).
//
// Integer / index work plus a little float32 geometry: every float expression below is written with ONE rounding per
// operation in the statement order of the reference sources (`#pragma clang fp contract(off)`: hipcc fuses a*b+c by
// default), so keep / pair indices are decided by the same float32 values the reference compares.
//
// An expanded box is (source row, zero mask, label): clean_class duplicates a box with one column of its class
// distribution set to 0, so a duplicate is fully described by the input row it came from and the set of zeroed columns.
#include <algorithm>

#include "kernels.h"

#pragma clang fp contract(off)

namespace sttran {

namespace {

constexpr int kOcMaxCols = 64;        // class-distribution columns (36 in the reference)
constexpr int kOcExpand = 4;           // copies one input box can have after clean_class(5), (8), (17): itself + one per pass,
                                      // each pass copying only the newest copy of a chain (label c -> c' -> c'')
constexpr int kOcMaxFrameBoxes = 1024; // expanded boxes per frame the NMS kernel holds in LDS (larger frames: global scratch)

// arg-max over the columns of a distribution row with the masked columns reading 0 (first maximum wins, like
// torch.argmax on distinct values)
__device__ __forceinline__ int argmax_masked(const float* __restrict__ row, int ncol, uint64_t zmask, int first, float* best_out) {
  int bi = first;
  float best = ((zmask >> first) & 1) ? 0.f : row[first];
  for (int c = first + 1; c < ncol; ++c) {
    const float v = ((zmask >> c) & 1) ? 0.f : row[c];
    if (v > best) { best = v; bi = c; }
  }
  if (best_out) *best_out = best;
  return bi;
}

// frame f owns input rows [fstart[f], fstart[f+1]) (boxes are sorted by frame id, as the detector emits them).
// `status` was written by objcls_check_order_kernel, launched in front of this kernel on the same stream: on unsorted
// input (value 2 set) the binary search below would give non-monotone ranges, so every frame is made EMPTY instead --
// the kernels behind this one (expand, NMS, row / pair writers) then loop over nothing, and the host returns
// STTRAN_ERR_ORDER without looking at their outputs (the Python wrapper sorts the rows and calls again).
__global__ void objcls_frame_ranges_kernel(const float* __restrict__ boxes, int B, int T, int* __restrict__ fstart,
                                           const int* __restrict__ status) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f > T) return;
  if (*status & 2) { fstart[f] = 0; return; }
  int lo = 0, hi = B;                       // first row whose frame id >= f
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (boxes[(int64_t)mid * 5] < (float)f) lo = mid + 1; else hi = mid;
  }
  fstart[f] = lo;
}

// The frame ranges above need the rows sorted by frame id, every id inside [0, T) (the reference selects rows with
// `boxes[:, 0] == i` and accepts any order, lib/sttran.py:59-62,205-207): checked here; status |= 2 (bit 1) otherwise
__global__ void objcls_check_order_kernel(const float* __restrict__ boxes, int B, int T, int* __restrict__ status) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const float f = boxes[(int64_t)r * 5];
  const bool bad = !(f >= 0.f) || !(f < (float)T) || f != floorf(f) || (r + 1 < B && boxes[(int64_t)(r + 1) * 5] < f);
  if (bad) atomicOr(status, 2);
}

// clean_class(5), clean_class(8), clean_class(17) of one frame per thread (lib/sttran.py:52-85,197-199): after each
// pass the frame's list is [what it was ..., a copy of every box whose label is the class, with that class's column
// zeroed and the label re-derived by arg-max].  Capacity 4 x the frame's input boxes: a pass copies an entry only if its
// label is the pass's class, and a copy's new label can only match a LATER pass, so a box has at most 1 + 3 entries.
__global__ void objcls_expand_kernel(const float* __restrict__ dist, const int64_t* __restrict__ labels, int ncol, int T,
                                     const int* __restrict__ fstart, int* __restrict__ ent_src, uint64_t* __restrict__ ent_mask,
                                     int* __restrict__ ent_label, int* __restrict__ n1) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  const int r0 = fstart[f], r1 = fstart[f + 1];
  const int64_t base = (int64_t)kOcExpand * r0;
  int n = 0;
  for (int r = r0; r < r1; ++r, ++n) {
    ent_src[base + n] = r; ent_mask[base + n] = 0; ent_label[base + n] = (int)labels[r];
  }
  const int classes[3] = {5, 8, 17};
  for (int s = 0; s < 3; ++s) {
    const int c = classes[s];
    if (c - 1 >= ncol) continue;
    const int n_before = n;
    for (int k = 0; k < n_before; ++k) {
      if (ent_label[base + k] != c) continue;
      const int src = ent_src[base + k];
      const uint64_t m = ent_mask[base + k] | (1ull << (c - 1));
      ent_src[base + n] = src; ent_mask[base + n] = m;
      ent_label[base + n] = argmax_masked(dist + (int64_t)src * ncol, ncol, m, 0, nullptr) + 1;
      ++n;
    }
  }
  n1[f] = n;
}

// IoU of two boxes whose extents count both border pixels (width = x2 - x1 + 1): the measure the reference's nms op
// thresholds (fasterRCNN/lib/model/csrc/cuda/nms.cu:13-21).  float32, one rounding per operation, union = area(p) +
// area(q) - overlap in that order -- the value decides keep / drop, so the operation order is part of the contract.
__device__ __forceinline__ float box_iou_inclusive(const float* p, const float* q) {
  const float ox = fmaxf(fminf(p[2], q[2]) - fmaxf(p[0], q[0]) + 1.f, 0.f);     // overlap extent in x
  const float oy = fmaxf(fminf(p[3], q[3]) - fmaxf(p[1], q[1]) + 1.f, 0.f);     // ... and in y
  const float overlap = ox * oy;
  const float area_p = (p[2] - p[0] + 1.f) * (p[3] - p[1] + 1.f);
  const float area_q = (q[2] - q[0] + 1.f) * (q[3] - q[1] + 1.f);
  return overlap / (area_p + area_q - overlap);
}

// One workgroup per frame: class of every expanded box = arg-max of its (masked) distribution; order by (class
// ascending, score descending, position ascending) = the order in which lib/sttran.py:211-236 emits the per-class
// groups; greedy suppression inside each class (nms.cu:96-118: a box is dropped when its IoU with an earlier kept box
// of the class exceeds the threshold; `ge` selects the CPU flavour's >=, nms_cpu.cpp:62).
// The frame's tables (box, score, class, order, suppressed) live in LDS for up to 1 024 expanded boxes -- every frame a
// detector with a few hundred proposals produces -- and in the caller's scratch for larger frames (same code, global
// memory: slower, not limited; the reference has no limit).
struct NmsTables {
  float (*bx)[4];
  float* score;
  int* cls;
  int* order;
  unsigned char* supp;
};
__device__ void nms_frame(const NmsTables& t_, const float* __restrict__ boxes, const float* __restrict__ dist, int ncol,
                          const int* __restrict__ ent_src, const uint64_t* __restrict__ ent_mask, int64_t base, int n, float thr,
                          int ge, int* __restrict__ kept, int* __restrict__ n2f) {
  const int tid = threadIdx.x;
  float (*bx)[4] = t_.bx; float* score = t_.score; int* cls = t_.cls; int* order = t_.order; unsigned char* supp = t_.supp;
  for (int t = tid; t < n; t += 256) {
    const int src = ent_src[base + t];
    float s;
    cls[t] = argmax_masked(dist + (int64_t)src * ncol, ncol, ent_mask[base + t], 0, &s);
    score[t] = s;
    for (int c = 0; c < 4; ++c) bx[t][c] = boxes[(int64_t)src * 5 + 1 + c];
    supp[t] = 0;
  }
  __syncthreads();
  for (int t = tid; t < n; t += 256) {
    const int ct = cls[t];
    const float st = score[t];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const int cu = cls[u];
      const float su = score[u];
      rank += (cu < ct) || (cu == ct && (su > st || (su == st && u < t)));
    }
    order[rank] = t;
  }
  __syncthreads();
  for (int p = 0; p < n; ++p) {
    if (!supp[p]) {                                   // workgroup-uniform: written before the last barrier
      const int ip = order[p];
      const int cp = cls[ip];
      for (int j = p + 1 + tid; j < n; j += 256) {
        const int ij = order[j];
        if (cls[ij] != cp) break;                       // sorted by class: the group has ended for this thread
        if (!supp[j]) {
          const float ovr = box_iou_inclusive(bx[ip], bx[ij]);
          if (ge ? (ovr >= thr) : (ovr > thr)) supp[j] = 1;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    int k = 0;
    for (int p = 0; p < n; ++p)
      if (!supp[p]) kept[base + k++] = order[p];
    *n2f = k;
  }
}

__global__ void __launch_bounds__(256)
objcls_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ dist, int ncol, const int* __restrict__ fstart,
                  const int* __restrict__ ent_src, const uint64_t* __restrict__ ent_mask, const int* __restrict__ n1,
                  float thr, int ge, int* __restrict__ kept, int* __restrict__ n2, float* __restrict__ big_f,
                  int* __restrict__ big_i, unsigned char* __restrict__ big_b) {
  __shared__ float bx[kOcMaxFrameBoxes][4];
  __shared__ float score[kOcMaxFrameBoxes];
  __shared__ int cls[kOcMaxFrameBoxes];
  __shared__ int order[kOcMaxFrameBoxes];
  __shared__ unsigned char supp[kOcMaxFrameBoxes];
  const int f = blockIdx.x;
  const int n = n1[f];
  const int64_t base = (int64_t)kOcExpand * fstart[f];
  NmsTables t;
  if (n <= kOcMaxFrameBoxes) {
    t = NmsTables{bx, score, cls, order, supp};
  } else {
    // this frame's slice of the caller's scratch (8 entries per input box, like ent_*): [box 4 | score 1] floats,
    // [class | order] ints, suppressed bytes
    t = NmsTables{reinterpret_cast<float (*)[4]>(big_f + 5 * base), big_f + 5 * base + 4 * (int64_t)n, big_i + 2 * base,
                  big_i + 2 * base + n, big_b + base};
  }
  nms_frame(t, boxes, dist, ncol, ent_src, ent_mask, base, n, thr, ge, kept, n2 + f);
}

// exclusive scan of per-frame counts by one thread (T is a few hundred at most) + the grand total
__global__ void objcls_scan_kernel(const int* __restrict__ cnt, int T, int* __restrict__ off, int* __restrict__ total) {
  if (blockIdx.x || threadIdx.x) return;
  int acc = 0;
  for (int f = 0; f < T; ++f) { off[f] = acc; acc += cnt[f]; }
  off[T] = acc;
  *total = acc;
}

// one workgroup per output row (grid = capacity; rows past the total exit): boxes, masked distribution, gathered
// features, source row, and pred_scores / pred_labels = max / arg-max over columns 1.. (+2)  (lib/sttran.py:239-244)
__global__ void __launch_bounds__(256)
objcls_write_rows_kernel(const float* __restrict__ boxes, const float* __restrict__ dist, const float* __restrict__ feats,
                         int ncol, int F, int T, const int* __restrict__ fstart, const int* __restrict__ ent_src,
                         const uint64_t* __restrict__ ent_mask, const int* __restrict__ kept, const int* __restrict__ off2,
                         float* __restrict__ o_boxes, float* __restrict__ o_dist, float* __restrict__ o_feats,
                         float* __restrict__ o_score, int64_t* __restrict__ o_label, int* __restrict__ o_src) {
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r >= off2[T]) return;
  int lo = 0, hi = T;                                  // frame f with off2[f] <= r < off2[f+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off2[mid] <= r) lo = mid; else hi = mid;
  }
  const int f = lo;
  const int64_t base = (int64_t)kOcExpand * fstart[f];
  const int e = kept[base + (r - off2[f])];
  const int src = ent_src[base + e];
  const uint64_t m = ent_mask[base + e];
  const float* drow = dist + (int64_t)src * ncol;
  if (tid < ncol) o_dist[(int64_t)r * ncol + tid] = ((m >> tid) & 1) ? 0.f : drow[tid];
  if (tid == 0) {
    o_boxes[(int64_t)r * 5] = (float)f;
    for (int c = 0; c < 4; ++c) o_boxes[(int64_t)r * 5 + 1 + c] = boxes[(int64_t)src * 5 + 1 + c];
    float best;
    const int bi = argmax_masked(drow, ncol, m, 1, &best);
    o_score[r] = best;
    o_label[r] = bi + 1;                                // column bi <-> class id bi + 1  (= arg-max over [:, 1:] + 2)
    if (o_src) o_src[r] = src;
  }
  if (o_feats) {
    const float* s = feats + (int64_t)src * F;
    float* d = o_feats + (int64_t)r * F;
    if ((F & 3) == 0) {
      for (int i = tid * 4; i < F; i += 1024) *reinterpret_cast<f32x4*>(d + i) = *reinterpret_cast<const f32x4*>(s + i);
    } else {
      for (int i = tid; i < F; i += 256) d[i] = s[i];
    }
  }
}

// HUMAN_IDX (lib/sttran.py:247-254): per frame the box with the highest person score (column 0), 0 for a frame without
// boxes -- and then, exactly like the reference's vectorised assignment, EVERY frame (empty ones included, whose index
// is 0) overwrites label and score of its human row.
__global__ void objcls_human_kernel(const float* __restrict__ o_dist, int ncol, int T, const int* __restrict__ off2,
                                    int64_t* __restrict__ human) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  int best_r = 0;
  float best = 0.f;
  for (int r = off2[f]; r < off2[f + 1]; ++r) {
    const float v = o_dist[(int64_t)r * ncol];
    if (r == off2[f] || v > best) { best = v; best_r = r; }
  }
  human[f] = best_r;
}
__global__ void objcls_override_kernel(const float* __restrict__ o_dist, int ncol, int T, const int64_t* __restrict__ human,
                                       int total_rows_hint, const int* __restrict__ off2, float* __restrict__ o_score,
                                       int64_t* __restrict__ o_label) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T || off2[T] == 0) return;
  const int64_t h = human[f];
  o_label[h] = 1;
  o_score[h] = o_dist[h * ncol];
}

// pairs (human of the frame, every box of the frame whose label is not 1), frame by frame (lib/sttran.py:256-266)
__global__ void objcls_pair_count_kernel(const int64_t* __restrict__ o_label, int T, const int* __restrict__ off2,
                                         int* __restrict__ np) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  int c = 0;
  for (int r = off2[f]; r < off2[f + 1]; ++r) c += o_label[r] != 1;
  np[f] = c;
}
__global__ void objcls_pair_write_kernel(const int64_t* __restrict__ o_label, int T, const int* __restrict__ off2,
                                         const int* __restrict__ poff, const int64_t* __restrict__ human,
                                         int64_t* __restrict__ pair_idx, float* __restrict__ im_idx) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  int p = poff[f];
  for (int r = off2[f]; r < off2[f + 1]; ++r) {
    if (o_label[r] == 1) continue;
    pair_idx[2 * (int64_t)p] = human[f];
    pair_idx[2 * (int64_t)p + 1] = r;
    im_idx[p] = (float)f;
    ++p;
  }
}

// ---- ROIAlign forward (what fasterRCNN/lib/model/csrc/cuda/ROIAlign_cuda.cu:65-118 computes) -------------------------
// A bilinear sample is separable: along each axis it is a pair of neighbouring cells with weights (1 - l, l), after the
// reference's clamping rules for that axis (a coordinate in [-1, 0] snaps to 0, one at or past the last cell snaps onto
// it, one outside [-1, size] makes the whole sample contribute 0).  The four corner weights are the products of the two
// axes' weights -- the same float32 products, in the same order, as the reference's w1..w4.
struct AxisTap {
  int lo, hi;
  float w_lo, w_hi;
  bool inside;
};
__device__ __forceinline__ AxisTap axis_tap(float v, int size) {
  AxisTap t;
  t.inside = !(v < -1.0f || v > (float)size);
  if (v <= 0) v = 0;
  t.lo = (int)v;
  if (t.lo >= size - 1) { t.hi = t.lo = size - 1; v = (float)t.lo; } else { t.hi = t.lo + 1; }
  t.w_hi = v - (float)t.lo;
  t.w_lo = 1.f - t.w_hi;
  return t;
}

// grid = (rois, channel slices); a workgroup owns one roi and a slice of channels, its threads walk (channel, bin) with
// the bin fastest.  The roi's geometry is derived once per thread, not once per output element.
__global__ void __launch_bounds__(256)
roi_align_kernel(const float* __restrict__ fmaps, float spatial_scale, int T, int channels, int height, int width, int pooled,
                 int sampling_ratio, const float* __restrict__ rois, float* __restrict__ out, int channels_per_block) {
  const int64_t n = blockIdx.x;
  const float* roi = rois + n * 5;
  const int frame = min(max((int)roi[0], 0), T - 1);       // (the reference would read out of bounds)
  // no rounding of the scaled corners; a malformed roi is forced to 1 x 1
  const float x0 = roi[1] * spatial_scale, y0 = roi[2] * spatial_scale;
  const float x1 = roi[3] * spatial_scale, y1 = roi[4] * spatial_scale;
  const float roi_w = fmaxf(x1 - x0, 1.f), roi_h = fmaxf(y1 - y0, 1.f);
  const float bin_h = roi_h / (float)pooled, bin_w = roi_w / (float)pooled;
  // adaptive sampling grid: ceil(roi extent / pooled) samples per bin and axis
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)pooled);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pooled);
  const float inv_count = (float)(gh * gw);
  const int bins = pooled * pooled;
  const int c_begin = blockIdx.y * channels_per_block, c_end = min(channels, c_begin + channels_per_block);
  for (int idx = threadIdx.x; idx < (c_end - c_begin) * bins; idx += 256) {
    const int c = c_begin + idx / bins, bin = idx % bins;
    const int ph = bin / pooled, pw = bin - ph * pooled;
    const float* map = fmaps + ((int64_t)frame * channels + c) * height * width;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      const AxisTap ty = axis_tap(y0 + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh, height);
      const float* row_lo = map + ty.lo * width;
      const float* row_hi = map + ty.hi * width;
      for (int ix = 0; ix < gw; ++ix) {
        const AxisTap tx = axis_tap(x0 + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw, width);
        float v = 0.f;
        if (ty.inside && tx.inside)
          v = (ty.w_lo * tx.w_lo) * row_lo[tx.lo] + (ty.w_lo * tx.w_hi) * row_lo[tx.hi] + (ty.w_hi * tx.w_lo) * row_hi[tx.lo] +
              (ty.w_hi * tx.w_hi) * row_hi[tx.hi];
        acc += v;
      }
    }
    out[(n * channels + c) * bins + bin] = acc / inv_count;
  }
}

inline char* carve(char*& p, size_t bytes) {
  char* r = p;
  p += (bytes + 255) & ~size_t(255);
  return r;
}

}  // namespace

size_t objcls_scratch_bytes(int64_t B, int T) {
  size_t n = 0;
  auto add = [&](size_t b) { n += (b + 255) & ~size_t(255); };
  add((size_t)(T + 2) * 4);                       // fstart
  add((size_t)kOcExpand * B * 4); add((size_t)kOcExpand * B * 8); add((size_t)kOcExpand * B * 4);   // ent_src / ent_mask / ent_label
  add((size_t)kOcExpand * B * 4);                         // kept
  add((size_t)kOcExpand * B * 5 * 4); add((size_t)kOcExpand * B * 2 * 4); add((size_t)kOcExpand * B);   // NMS tables of frames with more than 1 024 expanded boxes
  for (int i = 0; i < 5; ++i) add((size_t)(T + 2) * 4);   // n1, n2, off2, np, poff
  add(64);                                        // totals + status
  return n + 256;
}

// Everything up to the pair list.  Enqueues on `s`, then copies {rows, pairs, status} back and waits for it (the
// reference synchronises dozens of times in this branch; the sizes of the outputs are data dependent).
hipError_t launch_objcls_select(hipStream_t s, const float* boxes, const float* dist, const float* feats,
                                const int64_t* labels, int64_t B, int T, int ncol, int F, float thr, int ge, int64_t capacity,
                                float* o_boxes, float* o_dist, float* o_feats, float* o_score, int64_t* o_label, int* o_src,
                                int64_t* o_pair, float* o_im, int64_t* o_human, void* scratch, int32_t host_out[3]) {
  if (ncol > kOcMaxCols || ncol < 2 || B <= 0 || T <= 0 || capacity < 1) return hipErrorInvalidValue;
  char* p = reinterpret_cast<char*>(scratch);
  int* fstart = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* ent_src = reinterpret_cast<int*>(carve(p, (size_t)kOcExpand * B * 4));
  uint64_t* ent_mask = reinterpret_cast<uint64_t*>(carve(p, (size_t)kOcExpand * B * 8));
  int* ent_label = reinterpret_cast<int*>(carve(p, (size_t)kOcExpand * B * 4));
  int* kept = reinterpret_cast<int*>(carve(p, (size_t)kOcExpand * B * 4));
  float* big_f = reinterpret_cast<float*>(carve(p, (size_t)kOcExpand * B * 5 * 4));
  int* big_i = reinterpret_cast<int*>(carve(p, (size_t)kOcExpand * B * 2 * 4));
  unsigned char* big_b = reinterpret_cast<unsigned char*>(carve(p, (size_t)kOcExpand * B));
  int* n1 = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* n2 = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* off2 = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* np = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* poff = reinterpret_cast<int*>(carve(p, (size_t)(T + 2) * 4));
  int* totals = reinterpret_cast<int*>(carve(p, 64));    // [0] rows, [1] pairs, [2] status
  hipError_t e = hipMemsetAsync(totals, 0, 64, s);
  if (e != hipSuccess) return e;
  const int tb = 128, tg = (T + 1 + tb - 1) / tb;
  hipLaunchKernelGGL(objcls_check_order_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, boxes, (int)B, T, totals + 2);
  hipLaunchKernelGGL(objcls_frame_ranges_kernel, dim3(tg), dim3(tb), 0, s, boxes, (int)B, T, fstart, totals + 2);
  hipLaunchKernelGGL(objcls_expand_kernel, dim3(tg), dim3(tb), 0, s, dist, labels, ncol, T, fstart, ent_src, ent_mask, ent_label, n1);
  hipLaunchKernelGGL(objcls_nms_kernel, dim3(T), dim3(256), 0, s, boxes, dist, ncol, fstart, ent_src, ent_mask, n1, thr, ge, kept,
                     n2, big_f, big_i, big_b);
  hipLaunchKernelGGL(objcls_scan_kernel, dim3(1), dim3(64), 0, s, n2, T, off2, totals);
  const int64_t rows_cap = std::min<int64_t>(capacity, kOcExpand * B);
  hipLaunchKernelGGL(objcls_write_rows_kernel, dim3((unsigned)rows_cap), dim3(256), 0, s, boxes, dist, feats, ncol, F, T, fstart,
                     ent_src, ent_mask, kept, off2, o_boxes, o_dist, o_feats, o_score, o_label, o_src);
  hipLaunchKernelGGL(objcls_human_kernel, dim3(tg), dim3(tb), 0, s, o_dist, ncol, T, off2, o_human);
  hipLaunchKernelGGL(objcls_override_kernel, dim3(tg), dim3(tb), 0, s, o_dist, ncol, T, o_human, 0, off2, o_score, o_label);
  hipLaunchKernelGGL(objcls_pair_count_kernel, dim3(tg), dim3(tb), 0, s, o_label, T, off2, np);
  hipLaunchKernelGGL(objcls_scan_kernel, dim3(1), dim3(64), 0, s, np, T, poff, totals + 1);
  hipLaunchKernelGGL(objcls_pair_write_kernel, dim3(tg), dim3(tb), 0, s, o_label, T, off2, poff, o_human, o_pair, o_im);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(host_out, totals, 12, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

hipError_t launch_roi_align(hipStream_t s, const float* fmaps, int T, int C, int H, int W, const float* rois, int64_t P,
                            int pooled, float spatial_scale, int sampling_ratio, float* out) {
  if (P <= 0) return hipSuccess;
  if (P > 0x7fffffff) return hipErrorInvalidValue;
  // enough workgroups to fill the chip even for a handful of rois; >= 32 channels (x 49 bins) per workgroup
  const int want = std::max(1, (int)((int64_t)num_cus() * 8 / P));
  const int slices = std::min((C + 31) / 32, want);
  const int cpb = (C + slices - 1) / slices;
  hipLaunchKernelGGL(roi_align_kernel, dim3((unsigned)P, (unsigned)((C + cpb - 1) / cpb)), dim3(256), 0, s, fmaps, spatial_scale, T, C,
                     H, W, pooled, sampling_ratio, rois, out, cpb);
  return hipGetLastError();
}

}  // namespace sttran
