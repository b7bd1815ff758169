// kernels_front.hip -- pair-fusion front-end of STTran.forward (lib/sttran.py:381-399) and the
// sgdet ObjectClassifier input builder (lib/sttran.py:173-176).
#include "kernels.h"

namespace sttran {

// ------------------------------------------------------------------------------------------
// pair_prep: int64 pair_idx/labels -> int32 gather indices for the subj/obj FC GEMMs, and the two
// class-embedding blocks of rel_features: x[p, off:off+200] = E1[labels[subj]],
// x[p, off+200:off+400] = E2[labels[obj]]   (lib/sttran.py:390-396).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128)
pair_prep_kernel(const int64_t* __restrict__ pair_idx, const int64_t* __restrict__ labels, int P, int B,
                 int num_classes, const float* __restrict__ emb1, const float* __restrict__ emb2, int emb_dim,
                 int* __restrict__ subj_idx, int* __restrict__ obj_idx, float* __restrict__ x, int ldx,
                 int col_off, int* __restrict__ err_flag) {
  const int p = blockIdx.x;
  if (p >= P) return;
  int64_t s = pair_idx[2 * (int64_t)p], o = pair_idx[2 * (int64_t)p + 1];
  bool bad = (s < 0 || s >= B || o < 0 || o >= B);
  s = min(max(s, (int64_t)0), (int64_t)B - 1);
  o = min(max(o, (int64_t)0), (int64_t)B - 1);
  int64_t ls = labels[s], lo = labels[o];
  bad = bad || ls < 0 || ls >= num_classes || lo < 0 || lo >= num_classes;
  ls = min(max(ls, (int64_t)0), (int64_t)num_classes - 1);
  lo = min(max(lo, (int64_t)0), (int64_t)num_classes - 1);
  if (threadIdx.x == 0) {
    subj_idx[p] = (int)s;
    obj_idx[p] = (int)o;
    if (bad) atomicOr(err_flag, 1);
  }
  const int half = threadIdx.x >> 6, t = threadIdx.x & 63;
  const float* src = (half ? emb2 + lo * emb_dim : emb1 + ls * emb_dim);
  float* dst = x + (int64_t)p * ldx + col_off + half * emb_dim;
  for (int i = t * 4; i < emb_dim; i += 256) {
    *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
  }
}

hipError_t launch_pair_prep(hipStream_t s, const int64_t* pair_idx, const int64_t* labels, int P, int B,
                            int num_classes, const float* emb1, const float* emb2, int emb_dim, int* subj_idx,
                            int* obj_idx, float* x, int ldx, int col_off, int* err_flag) {
  hipLaunchKernelGGL(pair_prep_kernel, dim3(P), dim3(128), 0, s, pair_idx, labels, P, B, num_classes, emb1,
                     emb2, emb_dim, subj_idx, obj_idx, x, ldx, col_off, err_flag);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// mask_conv1_pool: Conv2d(2,128,k7,s2,p3) -> ReLU -> BatchNorm2d(eval) -> MaxPool2d(k3,s2,p1)
// (lib/sttran.py:338-341).  One workgroup per pair.  Thread <-> conv output position (14x14):
// its 2x7x7 receptive field lives in registers, the weights are broadcast from LDS as float4,
// four channels per step; the 14x14 map of those channels goes through LDS to be pooled 3x3/s2.
// ------------------------------------------------------------------------------------------
constexpr int kC1 = 128, kC1K = 98, kC1KP = 100;   // K padded to 100 (float4 rows)

__global__ void __launch_bounds__(256)
mask_conv1_pool_kernel(const float* __restrict__ masks, const float* __restrict__ w, const float* __restrict__ bias,
                       const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                       float* __restrict__ c2, int P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ws = smem;                       // [128][100]
  float* in = ws + kC1 * kC1KP;           // [2][33][33] zero-padded input
  float* cbuf = in + 2 * 33 * 33 + 2;     // [2][4][196] double-buffered conv maps (+2: keep 16B align)
  const int p = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < kC1 * kC1KP; i += 256) {
    const int c = i / kC1KP, k = i - c * kC1KP;
    ws[i] = (k < kC1K) ? w[c * kC1K + k] : 0.f;
  }
  for (int i = tid; i < 2 * 33 * 33; i += 256) {
    const int ci = i / 1089, r = i - ci * 1089, y = r / 33 - 3, x = r % 33 - 3;
    in[i] = (y >= 0 && y < 27 && x >= 0 && x < 27) ? masks[((int64_t)p * 2 + ci) * 729 + y * 27 + x] : 0.f;
  }
  __syncthreads();
  const bool active = tid < 196;
  const int oy = active ? tid / 14 : 0, ox = active ? tid % 14 : 0;
  float rf[kC1KP];
#pragma unroll
  for (int k = 0; k < kC1KP; ++k) {
    if (k < kC1K) {
      const int ci = k / 49, ky = (k % 49) / 7, kx = k % 7;
      rf[k] = in[ci * 1089 + (2 * oy + ky) * 33 + 2 * ox + kx];
    } else rf[k] = 0.f;
  }
  // pooling role of this thread: (channel-in-group, pooled position)
  const int pc = tid / 49, pp = tid % 49, py = pp / 7, px = pp % 7;
  for (int c0 = 0; c0 < kC1; c0 += 4) {
    float* cb = cbuf + ((c0 >> 2) & 1) * 4 * 196;
    if (active) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = c0 + cc;
        const f32x4* wr = reinterpret_cast<const f32x4*>(ws + c * kC1KP);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int q = 0; q < kC1KP / 4; ++q) {
          const f32x4 wv = wr[q];
          a0 = fmaf(wv[0], rf[4 * q + 0], a0);
          a1 = fmaf(wv[1], rf[4 * q + 1], a1);
          a0 = fmaf(wv[2], rf[4 * q + 2], a0);
          a1 = fmaf(wv[3], rf[4 * q + 3], a1);
        }
        float v = fmaxf(a0 + a1 + bias[c], 0.f);
        cb[cc * 196 + tid] = v * bn_scale[c] + bn_shift[c];
      }
    }
    __syncthreads();
    if (tid < 196) {
      float m = -INFINITY;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int y = 2 * py + dy, x = 2 * px + dx;
          if (y >= 0 && y < 14 && x >= 0 && x < 14) m = fmaxf(m, cb[pc * 196 + y * 14 + x]);
        }
      c2[((int64_t)p * kC1 + c0 + pc) * 49 + pp] = m;
    }
    // no second barrier: the next step writes the other half of cbuf; the barrier of that step
    // orders it against this step's reads before this half is written again.
  }
}

hipError_t launch_mask_conv1_pool(hipStream_t s, const float* masks, const float* w, const float* bias,
                                  const float* bn_scale, const float* bn_shift, float* c2, int P) {
  const size_t lds = (size_t)(kC1 * kC1KP + 2 * 33 * 33 + 2 + 2 * 4 * 196) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mask_conv1_pool_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  hipLaunchKernelGGL(mask_conv1_pool_kernel, dim3(P), dim3(256), lds, s, masks, w, bias, bn_scale, bn_shift,
                     c2, P);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// im2col for Conv2d(128,256,k3,p1) on [P,128,7,7]: row (p,hw), column ci*9 + ky*3 + kx -- the
// order of conv.4.weight.view(256, 1152) (lib/sttran.py:342).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
im2col3x3_kernel(const float* __restrict__ c2, float* __restrict__ cols, int P) {
  __shared__ float t[128 * 49];
  const int p = blockIdx.x;
  for (int i = threadIdx.x; i < 128 * 49; i += 256) t[i] = c2[(int64_t)p * 128 * 49 + i];
  __syncthreads();
  float* dst = cols + (int64_t)p * 49 * 1152;
  for (int i = threadIdx.x; i < 49 * 1152; i += 256) {
    const int hw = i / 1152, k = i - hw * 1152;
    const int ci = k / 9, kk = k - ci * 9, ky = kk / 3, kx = kk - ky * 3;
    const int y = hw / 7 + ky - 1, x = hw % 7 + kx - 1;
    dst[i] = (y >= 0 && y < 7 && x >= 0 && x < 7) ? t[ci * 49 + y * 7 + x] : 0.f;
  }
}

hipError_t launch_im2col3x3(hipStream_t s, const float* c2, float* cols, int P) {
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(P), dim3(256), 0, s, c2, cols, P);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// objcls_prep: ObjectClassifier input of the sgdet+wks branch (lib/sttran.py:174-176):
//   z[b] = [ features[b] | distribution[b] @ obj_embed.weight | ReLU(Linear(BN1d(center_size(box)))) ]
// center_size (lib/fpn/box_utils.py:51-63): wh = xy2 - xy1 + 1, c = xy1 + 0.5 wh.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
objcls_prep_kernel(const float* __restrict__ features, const float* __restrict__ dist,
                   const float* __restrict__ boxes, const float* __restrict__ E0,
                   const float* __restrict__ pos_scale, const float* __restrict__ pos_shift,
                   const float* __restrict__ pos_w, const float* __restrict__ pos_b, float* __restrict__ z, int B,
                   int feat_dim, int ncls, int emb_dim) {
  __shared__ float d[64];
  __shared__ float cs[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int zd = feat_dim + emb_dim + 128;
  float* zr = z + (int64_t)b * zd;
  if (tid < ncls) d[tid] = dist[(int64_t)b * ncls + tid];
  if (tid == 0) {
    const float* bx = boxes + (int64_t)b * 5 + 1;
    const float w = bx[2] - bx[0] + 1.0f, h = bx[3] - bx[1] + 1.0f;
    const float c[4] = {bx[0] + 0.5f * w, bx[1] + 0.5f * h, w, h};
    for (int i = 0; i < 4; ++i) cs[i] = c[i] * pos_scale[i] + pos_shift[i];
  }
  for (int i = tid * 4; i < feat_dim; i += 1024)
    *reinterpret_cast<f32x4*>(zr + i) = *reinterpret_cast<const f32x4*>(features + (int64_t)b * feat_dim + i);
  __syncthreads();
  for (int j = tid; j < emb_dim; j += 256) {
    float a = 0.f;
    for (int c = 0; c < ncls; ++c) a = fmaf(d[c], E0[c * emb_dim + j], a);
    zr[feat_dim + j] = a;
  }
  if (tid < 128) {
    float a = pos_b[tid];
    for (int i = 0; i < 4; ++i) a = fmaf(pos_w[tid * 4 + i], cs[i], a);
    zr[feat_dim + emb_dim + tid] = fmaxf(a, 0.f);
  }
}

hipError_t launch_objcls_prep(hipStream_t s, const float* features, const float* dist, const float* boxes,
                              const float* E0, const float* pos_scale, const float* pos_shift, const float* pos_w,
                              const float* pos_b, float* z, int B, int feat_dim, int ncls, int emb_dim) {
  if (ncls > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(objcls_prep_kernel, dim3(B), dim3(256), 0, s, features, dist, boxes, E0, pos_scale,
                     pos_shift, pos_w, pos_b, z, B, feat_dim, ncls, emb_dim);
  return hipGetLastError();
}

}  // namespace sttran
