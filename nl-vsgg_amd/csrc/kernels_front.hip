// kernels_front.hip -- pair-fusion front-end of STTran.forward (lib/sttran.py:381-399) and the
// sgdet ObjectClassifier input builder (lib/sttran.py:173-176).
#include <algorithm>

#include "kernels.h"

namespace sttran {

// ------------------------------------------------------------------------------------------
// pair_prep: where every pair's inputs live, and the two class-embedding blocks of rel_features.
//
// The inputs of a call are a list of CHUNKS (ChunkTable, kernels.h): one chunk = one clip's tensors where the caller
// left them (SttranInputs' per-clip pointer tables), or a single chunk holding the whole contiguous batch.  Per pair p
// (chunk c, local pair lp, local box rows s / o from the chunk's own pair_idx):
//   feat_off[0][p], feat_off[1][p] = element offset of the subject / object feature row from chunk 0's features
//                                    (the A-operand row gather of the subj / obj FC GEMMs, GemmOperand::rowoff)
//   union_off[p], mask_off[p]      = element offset of the pair's union_feat / spatial_masks block from chunk 0's
//   x[p, off:off+200] = E1[labels[subj]],  x[p, off+200:off+400] = E2[labels[obj]]     (lib/sttran.py:390-396)
//   cls_of_pair / subj_of_pair     = (DSG-DETR) class of the object box, GLOBAL row of the subject box
// Out-of-range pair_idx / labels are clamped and flagged (err_flag bit 0), as before.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int chunk_of(const int64_t* __restrict__ start, int n, int64_t v) {
  int lo = 0, hi = n;                       // last chunk c with start[c] <= v (empty chunks are skipped that way)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (start[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t float_distance(const void* a, const void* b) {
  return (reinterpret_cast<intptr_t>(a) - reinterpret_cast<intptr_t>(b)) / 4;
}

__global__ void __launch_bounds__(128)
pair_prep_kernel(ChunkTable tab, int P, int feat_dim, int num_classes, const float* __restrict__ emb1,
                 const float* __restrict__ emb2, int emb_dim, int64_t* __restrict__ feat_off, int64_t* __restrict__ union_off,
                 int64_t* __restrict__ mask_off, int* __restrict__ cls_of_pair, int* __restrict__ subj_of_pair,
                 float* __restrict__ x, int ldx, int col_off, int* __restrict__ err_flag) {
  const int p = blockIdx.x;
  if (p >= P) return;
  const int c = chunk_of(tab.pair_start, tab.n, p);
  const int64_t lp = p - tab.pair_start[c];
  const int64_t B = tab.box_start[c + 1] - tab.box_start[c];
  const int64_t* pair_idx = reinterpret_cast<const int64_t*>(tab.pair_idx[c]);
  const int64_t* labels = reinterpret_cast<const int64_t*>(tab.labels[c]);
  int64_t s = pair_idx[2 * lp], o = pair_idx[2 * lp + 1];
  bool bad = (s < 0 || s >= B || o < 0 || o >= B);
  s = min(max(s, (int64_t)0), B - 1);
  o = min(max(o, (int64_t)0), B - 1);
  int64_t ls = labels[s], lo = labels[o];
  bad = bad || ls < 0 || ls >= num_classes || lo < 0 || lo >= num_classes;
  ls = min(max(ls, (int64_t)0), (int64_t)num_classes - 1);
  lo = min(max(lo, (int64_t)0), (int64_t)num_classes - 1);
  if (threadIdx.x == 0) {
    const int64_t f0 = float_distance(tab.features[c], tab.features[tab.base]);
    feat_off[p] = f0 + s * feat_dim;
    feat_off[(int64_t)P + p] = f0 + o * feat_dim;
    union_off[p] = float_distance(tab.union_feat[c], tab.union_feat[tab.base]) + lp * ((int64_t)feat_dim * 49);
    mask_off[p] = float_distance(tab.masks[c], tab.masks[tab.base]) + lp * 1458;
    if (cls_of_pair) { cls_of_pair[p] = (int)lo; subj_of_pair[p] = (int)(tab.box_start[c] + s); }
    if (bad) atomicOr(err_flag, 1);
  }
  const int half = threadIdx.x >> 6, t = threadIdx.x & 63;
  const float* src = (half ? emb2 + lo * emb_dim : emb1 + ls * emb_dim);
  float* dst = x + (int64_t)p * ldx + col_off + half * emb_dim;
  for (int i = t * 4; i < emb_dim; i += 256) {
    *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
  }
}

hipError_t launch_pair_prep(hipStream_t s, const ChunkTable& tab, int P, int feat_dim, int num_classes, const float* emb1,
                            const float* emb2, int emb_dim, int64_t* feat_off, int64_t* union_off, int64_t* mask_off,
                            int* cls_of_pair, int* subj_of_pair, float* x, int ldx, int col_off, int* err_flag) {
  hipLaunchKernelGGL(pair_prep_kernel, dim3(P), dim3(128), 0, s, tab, P, feat_dim, num_classes, emb1, emb2, emb_dim,
                     feat_off, union_off, mask_off, cls_of_pair, subj_of_pair, x, ldx, col_off, err_flag);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// union_boxes_masks: the step right before the hot path (lib/object_detector.py:110-124), which the
// reference does with a device->host copy, a Cython loop (lib/draw_rectangles/draw_rectangles.pyx:27-67)
// and a host->device copy.  Per pair: the union box of subject and object, and the two soft box masks
//   mask[i][y][x] = clamp01(x+1-x1) * clamp01(x2-x) * clamp01(y+1-y1) * clamp01(y2-y)  - 0.5
// with the box rescaled into the union box on a pool x pool grid (float32, same operation order).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
union_boxes_masks_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ pair_idx,
                         const float* __restrict__ im_idx, int P, int pool, float* __restrict__ union_boxes,
                         float* __restrict__ masks) {
  const int p = blockIdx.x;
  if (p >= P) return;
  __shared__ float bx[8];
  __shared__ float sc[2][4];      // per box: x1, y1, x2, y2 in pooled coordinates
  if (threadIdx.x < 8) {
    const int64_t b = pair_idx[2 * (int64_t)p + (threadIdx.x >> 2)];
    bx[threadIdx.x] = boxes[b * 5 + 1 + (threadIdx.x & 3)];
  }
  __syncthreads();
  const float x1u = fminf(bx[0], bx[4]), y1u = fminf(bx[1], bx[5]);
  const float x2u = fmaxf(bx[2], bx[6]), y2u = fmaxf(bx[3], bx[7]);
  if (threadIdx.x < 2) {
    const int i = threadIdx.x;
    const float w = x2u - x1u, h = y2u - y1u, ps = (float)pool;
    sc[i][0] = (bx[0 + 4 * i] - x1u) * ps / w;
    sc[i][1] = (bx[1 + 4 * i] - y1u) * ps / h;
    sc[i][2] = (bx[2 + 4 * i] - x1u) * ps / w;
    sc[i][3] = (bx[3 + 4 * i] - y1u) * ps / h;
  }
  if (union_boxes && threadIdx.x == 0) {
    float* u = union_boxes + (int64_t)p * 5;
    u[0] = im_idx ? im_idx[p] : 0.f; u[1] = x1u; u[2] = y1u; u[3] = x2u; u[4] = y2u;
  }
  __syncthreads();
  const int n = pool * pool;
  for (int t = threadIdx.x; t < 2 * n; t += 256) {
    const int i = t / n, r = t - i * n, y = r / pool, x = r - y * pool;
    const float yc = fminf(fmaxf((float)(y + 1) - sc[i][1], 0.f), 1.f) * fminf(fmaxf(sc[i][3] - (float)y, 0.f), 1.f);
    const float xc = fminf(fmaxf((float)(x + 1) - sc[i][0], 0.f), 1.f) * fminf(fmaxf(sc[i][2] - (float)x, 0.f), 1.f);
    masks[(int64_t)p * 2 * n + t] = xc * yc - 0.5f;
  }
}

hipError_t launch_union_boxes_masks(hipStream_t s, const float* boxes, const int64_t* pair_idx, const float* im_idx,
                                    int P, int pool, float* union_boxes, float* masks) {
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(union_boxes_masks_kernel, dim3(P), dim3(256), 0, s, boxes, pair_idx, im_idx, P, pool,
                     union_boxes, masks);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// objcls_prep: ObjectClassifier input of the sgdet+wks branch (lib/sttran.py:174-176):
//   z[b] = [ features[b] | distribution[b] @ obj_embed.weight | ReLU(Linear(BN1d(center_size(box)))) ]
// center_size (lib/fpn/box_utils.py:51-63): wh = xy2 - xy1 + 1, c = xy1 + 0.5 wh.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
objcls_prep_kernel(ChunkTable tab, const float* __restrict__ E0,
                   const float* __restrict__ pos_scale, const float* __restrict__ pos_shift,
                   const float* __restrict__ pos_w, const float* __restrict__ pos_b, float* __restrict__ z,
                   int64_t ldz, int B, int feat_dim, int ncls, int emb_dim) {
  __shared__ float d[64];
  __shared__ float cs[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  // box row b of the batch = row lb of chunk c (one chunk per clip when the caller passed pointer tables)
  const int c = chunk_of(tab.box_start, tab.n, b);
  const int64_t lb = b - tab.box_start[c];
  const float* features = reinterpret_cast<const float*>(tab.features[c]) + lb * feat_dim;
  const float* dist = reinterpret_cast<const float*>(tab.dist[c]) + lb * ncls;
  const float* boxes = reinterpret_cast<const float*>(tab.boxes[c]) + lb * 5;
  float* zr = z + (int64_t)b * ldz;
  if (tid < ncls) d[tid] = dist[tid];
  if (tid == 0) {
    const float* bx = boxes + 1;
    const float w = bx[2] - bx[0] + 1.0f, h = bx[3] - bx[1] + 1.0f;
    const float c[4] = {bx[0] + 0.5f * w, bx[1] + 0.5f * h, w, h};
    for (int i = 0; i < 4; ++i) cs[i] = c[i] * pos_scale[i] + pos_shift[i];
  }
  for (int i = tid * 4; i < feat_dim; i += 1024)
    *reinterpret_cast<f32x4*>(zr + i) = *reinterpret_cast<const f32x4*>(features + i);
  __syncthreads();
  for (int j = tid; j < emb_dim; j += 256) {
    float a = 0.f;
    for (int c = 0; c < ncls; ++c) a = fmaf(d[c], E0[c * emb_dim + j], a);
    zr[feat_dim + j] = a;
  }
  if (tid < 128) {
    float a = pos_b[tid];
    for (int i = 0; i < 4; ++i) a = fmaf(pos_w[tid * 4 + i], cs[i], a);
    zr[feat_dim + emb_dim + tid] = relu_nan(a);
  }
}

hipError_t launch_objcls_prep(hipStream_t s, const ChunkTable& tab, const float* E0, const float* pos_scale,
                              const float* pos_shift, const float* pos_w, const float* pos_b, float* z, int64_t ldz, int B,
                              int feat_dim, int ncls, int emb_dim) {
  if (ncls > 64 || ldz < feat_dim + emb_dim + 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(objcls_prep_kernel, dim3(B), dim3(256), 0, s, tab, E0, pos_scale,
                     pos_shift, pos_w, pos_b, z, ldz, B, feat_dim, ncls, emb_dim);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// DSG-DETR class sequences on the device (lib/dsg_detr.py:545-555), one workgroup per clip.
//   sequence slot  = clip * NC + class (empty slots have length 0)            -> dec_off / dec_len
//   token order    = the clip's pairs, stably grouped by the class of their object box -> dec_src (token -> pair),
//                    out_src (pair -> P + token)
//   position index = handed out BY POSITION like the reference does (`[0]*count_0 + [1]*count_1 + ...` over the sorted
//                    unique subject boxes): token i of a sequence gets the dense rank of the i-th SMALLEST subject -> need
// Quadratic in the pairs of a clip (rank by counting), which is a few thousand at most; no host read-back.
// err_flag: bit 0 = pair_idx / labels out of range (clamped), bit 1 = more position indices than PE rows (clamped) or a
// sequence longer than max_len (the attention's key limit: such a sequence is skipped).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dsg_layout_kernel(const int64_t* __restrict__ pair_idx, const int64_t* __restrict__ labels, int B, const int* __restrict__ clip_start,
                  int NC, int P, int pe_rows, int max_len, int* __restrict__ dec_off, int* __restrict__ dec_len, int* __restrict__ dec_src,
                  int* __restrict__ need, int* __restrict__ out_src, int* cls_of_pair, int* subj_of_pair, int* subj_of_tok,
                  int* first_of_tok, int* err_flag, int lds_ints) {
  // pair_idx == nullptr: cls_of_pair / subj_of_pair were filled (and range-checked) by pair_prep_kernel already -- the
  // forward's path, where the pairs of a batch may live in per-clip tensors; otherwise they are derived here
  // with room in LDS (2 ints per pair of the largest clip) the subject / first-occurrence tables of the clip live there
  // (rows s .. s+n of the global scratch otherwise): the two passes below read them O(sequence length) times per token
  extern __shared__ int dyn[];
  __shared__ int hist[64], off[64];
  const int c = blockIdx.x, s = clip_start[c], n = clip_start[c + 1] - s, tid = threadIdx.x;
  const bool in_lds = 2 * n <= lds_ints;
  // both tables are indexed by (token - s); no pointer is ever formed outside its buffer (an LDS pointer minus s wraps in
  // its 32-bit address space and faults once it is used as a flat address)
  int* subj_t = in_lds ? dyn : subj_of_tok + s;
  int* first_t = in_lds ? dyn + n : first_of_tok + s;
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    if (!pair_idx) { atomicAdd(&hist[cls_of_pair[s + i]], 1); continue; }
    int64_t sj = pair_idx[2 * (int64_t)(s + i)], ob = pair_idx[2 * (int64_t)(s + i) + 1];
    bool bad = sj < 0 || sj >= B || ob < 0 || ob >= B;
    sj = sj < 0 ? 0 : (sj >= B ? B - 1 : sj);
    ob = ob < 0 ? 0 : (ob >= B ? B - 1 : ob);
    int64_t lab = labels[ob];
    bad |= lab < 0 || lab >= NC;
    lab = lab < 0 ? 0 : (lab >= NC ? NC - 1 : lab);
    if (bad) atomicOr(err_flag, 1);
    cls_of_pair[s + i] = (int)lab;
    subj_of_pair[s + i] = (int)sj;
    atomicAdd(&hist[lab], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int k = 0; k < NC; ++k) { off[k] = a; a += hist[k]; }
  }
  __syncthreads();
  if (tid < NC) {
    dec_off[c * NC + tid] = s + off[tid]; dec_len[c * NC + tid] = hist[tid];
    if (hist[tid] > max_len) atomicOr(err_flag, 2);      // the attention skips such a sequence
  }
  // stable position of pair i inside its class = the pairs before it with the same class: the clip is walked in chunks
  // of 256 pairs, `placed` counts per class what the earlier chunks held, the rank inside the chunk is counted from an
  // LDS copy of the chunk's classes (every lane reads the same byte: a broadcast)
  __shared__ int placed[64];
  __shared__ unsigned char chunk_cls[256];
  if (tid < 64) placed[tid] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + tid;
    const int ci = i < n ? cls_of_pair[s + i] : 255;
    chunk_cls[tid] = (unsigned char)ci;
    __syncthreads();
    if (i < n) {
      int r = placed[ci];
      for (int j = 0; j < tid; ++j) r += chunk_cls[j] == ci;
      const int tok = s + off[ci] + r;
      dec_src[tok] = s + i;
      out_src[s + i] = P + tok;
      subj_t[tok - s] = subj_of_pair[s + i];
    }
    __syncthreads();
    if (i < n) atomicAdd(&placed[ci], 1);
    __syncthreads();
  }
  // first occurrence of its subject inside the sequence?
  for (int i = tid; i < n; i += 256) {
    const int t = s + i, ci = cls_of_pair[dec_src[t]], o = s + off[ci], sv = subj_t[i];
    int f = 1;
    for (int k = o; k < t; ++k) f &= subj_t[k - s] != sv;
    first_t[i] = f;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int t = s + i, ci = cls_of_pair[dec_src[t]], o = s + off[ci], len = hist[ci], sv = subj_t[i];
    int r = 0, d = 0;                       // r = position of this token's subject in the sorted sequence (stable), d = its dense rank
    for (int k = o; k < o + len; ++k) {
      const int sk = subj_t[k - s];
      r += (sk < sv) || (sk == sv && k < t);
      d += (sk < sv) && first_t[k - s];
    }
    if (d >= pe_rows) { d = pe_rows - 1; atomicOr(err_flag, 2); }
    need[o + r] = d;
  }
}

hipError_t launch_dsg_layout(hipStream_t s, const int64_t* pair_idx, const int64_t* labels, int B, const int* clip_start,
                             int num_clips, int NC, int P, int pe_rows, int max_len, int* dec_off, int* dec_len, int* dec_src,
                             int* need, int* out_src, int* scratch4p, int* err_flag, int max_clip_pairs) {
  if (num_clips <= 0 || P <= 0) return hipSuccess;
  if (NC > 64) return hipErrorInvalidValue;
  // LDS for the largest clip's two per-token tables when they fit 48 KB (6 144 pairs), none otherwise
  const int lds_ints = (max_clip_pairs > 0 && max_clip_pairs <= 6144) ? 2 * max_clip_pairs : 0;
  hipLaunchKernelGGL(dsg_layout_kernel, dim3(num_clips), dim3(256), lds_ints * 4, s, pair_idx, labels, B, clip_start, NC, P, pe_rows,
                     max_len, dec_off, dec_len, dec_src, need, out_src, scratch4p, scratch4p + P, scratch4p + 2 * (int64_t)P,
                     scratch4p + 3 * (int64_t)P, err_flag, lds_ints);
  return hipGetLastError();
}

}  // namespace sttran
