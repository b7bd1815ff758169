// kernels_gemm_x3.hip -- launchers of the bf16x3 fp32-emulation GEMM (gemm_bf16x3.h; experiment, opt-in)
#include "gemm_bf16x3.h"
#include "gemm_launch.h"

namespace sttran {

hipError_t split_planes(hipStream_t s, const float* W, int64_t ld, int rows, int cols, void* planes, int64_t ldp) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_planes_kernel, dim3(rows), dim3(256), 0, s, W, ld, rows, cols, reinterpret_cast<__bf16*>(planes), ldp);
  return hipGetLastError();
}

template <class T, class Epi>
static hipError_t launch_x3(hipStream_t s, const GemmOperand& A, const X3Weights& B, int M, int N, int K, float* slab, const Epi& epi) {
  static DeviceMarks marks;
  auto kern = gemm_x3_kernel<T, Epi>;
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int tm = (M + T::BM - 1) / T::BM, tn = (N + T::BN - 1) / T::BN, tiles = tm * tn;
  const int ksteps = (K + kBK - 1) / kBK;
  const SkPlan sp = sk_plan(TILE_256x128, tiles, ksteps);           // one workgroup per CU (147 KB of LDS)
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, tm, tiles, ksteps, sp.dp_per_wg, sp.g_sk, base,
                     rem, slab, epi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !split) return e;
  hipLaunchKernelGGL((gemm_fixup_vec_kernel<T, Epi>), dim3(sp.tiles_sk, T::TM * T::TN * 4), dim3(T::NT), 0, s, M, N, tm, tn, ksteps,
                     sp.g_sk, base, rem, tiles - sp.tiles_sk, slab, epi);
  return hipGetLastError();
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// planes: [3][N][ldp] bf16 of the weight matrix (split_planes); A fp32 with the padded-operand contract (readable and
// finite up to ceil32(K) columns per row)
hipError_t gemm_linear_x3(hipStream_t s, const GemmOperand& A, const void* planes, int64_t ldp, int64_t plane_stride, int M, int N,
                          int K, const EpiLinear& epi, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if ((ldp & 31) || ldp < ((K + 31) & ~31) || !al16(planes) || !al16(A.ptr) || (A.ld & 3)) return hipErrorInvalidValue;
  X3Weights B{reinterpret_cast<const __bf16*>(planes), ldp, plane_stride};
  using T = X3Tile<256, 128, 4, 2>;
  const bool vec = (N & 3) == 0 && al16(epi.C) && (epi.ldc & 3) == 0 && al16(epi.bias) && al16(epi.rowbias) && (epi.rb_ld & 3) == 0 &&
                   (epi.rb_cols & 3) == 0 && al16(epi.scale) && al16(epi.shift) && al16(epi.res) && (epi.ldres & 3) == 0;
  if (vec) return launch_x3<T, EpiLinearV>(s, A, B, M, N, K, slab, EpiLinearV{epi});
  return launch_x3<T, EpiScalar4<EpiLinear>>(s, A, B, M, N, K, slab, EpiScalar4<EpiLinear>{epi});
}

// union_func1 on the bf16x3 engine: M = 49 P rows (pair, hw) read in place from the NCHW tensor (A_UNION_FLAT),
// N = 256 out channels from the pre-split weight planes [3][256][K], all of them in one 128x256 tile (U is staged once)
hipError_t launch_union_conv_x3(hipStream_t s, const float* U, const int64_t* u_off, const void* planes, const float* bias,
                                float* V, int P, int K, float* slab) {
  if (K % kBK != 0 || P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30) || !al16(planes)) return hipErrorInvalidValue;
  GemmOperand A{U, (int64_t)K * kUHW, nullptr, P, u_off};
  X3Weights B{reinterpret_cast<const __bf16*>(planes), (int64_t)K, (int64_t)256 * K};
  EpiUnionRows epi{V, bias, 256};
  return launch_x3<X3Tile<128, 256, 2, 4, A_UNION_FLAT>, EpiUnionRows>(s, A, B, P * kUHW, 256, K, slab, epi);
}

// Conv2d(128,256,k3,p1) -> ReLU -> BN on the same engine: planes = [3][256][1152] bf16 of the (ky, kx, ci)-ordered weight,
// c2 = channel-last [P][7][7][128]
hipError_t launch_mask_conv2_x3(hipStream_t s, const void* planes, const float* c2, const float* bias, const float* scale,
                                const float* shift, float* V, int P, float* slab) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30) || !al16(planes) || !al16(c2)) return hipErrorInvalidValue;
  GemmOperand A{c2, 0, nullptr, 0};
  X3Weights B{reinterpret_cast<const __bf16*>(planes), 1152, (int64_t)256 * 1152};
  EpiConvRows epi{V, bias, scale, shift, 256};
  return launch_x3<X3Tile<128, 256, 2, 4, A_CONV2>, EpiConvRows>(s, A, B, P * kUHW, 256, 1152, slab, epi);
}

}  // namespace sttran
