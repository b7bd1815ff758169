// kernels_gemm_t16c.hip -- union 1x1 conv and conv3x3 on the 16x16x4 kernel structure (gemm_f32_t16c.h): launchers
#include "gemm_f32_t16c.h"
#include "gemm_launch.h"

namespace sttran {

template <class T, class Epi>
static hipError_t launch_t16c(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K, float* slab,
                              const Epi& epi, int tile_base) {
  static DeviceMarks marks;
  auto kern = gemm16c_kernel<T, Epi>;
#ifdef STTRAN_GEMM_EXPERIMENT
  {
    static DeviceMarks m[9];
    const int abl = exp_env("STTRAN_T16C_ABLATE") ? atoi(exp_env("STTRAN_T16C_ABLATE")) : 0;
    switch (abl) {
      case 1: kern = gemm16c_kernel<T, Epi, 1>; break;
      case 2: kern = gemm16c_kernel<T, Epi, 2>; break;
      case 3: kern = gemm16c_kernel<T, Epi, 3>; break;
      case 4: kern = gemm16c_kernel<T, Epi, 4>; break;
      case 5: kern = gemm16c_kernel<T, Epi, 5>; break;
      case 6: kern = gemm16c_kernel<T, Epi, 6>; break;
      case 7: kern = gemm16c_kernel<T, Epi, 7>; break;
      case 8: kern = gemm16c_kernel<T, Epi, 8>; break;
      default: break;
    }
    if (abl >= 1 && abl <= 8 && m[abl].raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown;
  }
#endif
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  if (M != T::BM || K % kBK != 0 || N <= 0) return hipErrorInvalidValue;
  const int tm = 1, tn = (N + T::BN - 1) / T::BN, tiles_all = tm * tn;
  if (tile_base < 0 || tile_base >= tiles_all) return hipErrorInvalidValue;
  const int tiles = tiles_all - tile_base;                           // tile_base > 0: the tiles behind the fused launch's
  const int ksteps = K / kBK;
  const SkPlan sp = sk_plan(TILE_256x128, tiles, ksteps);            // one workgroup per CU (96 KB of LDS), 256 x 128 park slots
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, tm, tiles_all, ksteps, sp.dp_per_wg, sp.g_sk,
                     base, rem, tile_base, slab, epi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !split) return e;
  hipLaunchKernelGGL((gemm16c_fixup_kernel<T, Epi>), dim3(sp.tiles_sk, 2 * T::NB), dim3(T::NT), 0, s, M, N, tm, tn, ksteps,
                     sp.g_sk, base, rem, tile_base + tiles - sp.tiles_sk, slab, epi);
  return hipGetLastError();
}

// union_func1: V[p][c][hw] += W[c][:] . U[p][:][hw] + b[c]  (u_off optional: pair p's block starts at U + u_off[p] floats)
hipError_t launch_union_conv_t16(hipStream_t s, const float* U, const int64_t* u_off, const float* W, const float* bias, float* V,
                                 int P, int K, float* slab, int tile_base) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
  GemmOperand A{W, (int64_t)K, nullptr, 0, nullptr};
  GemmOperand B{U, (int64_t)K * kUHW, nullptr, P, u_off};
  return launch_t16c<Tile16C<B_UNION_FLAT>, EpiUnionT16>(s, A, B, 256, P * kUHW, K, slab, EpiUnionT16{V, bias, 256}, tile_base);
}

// Conv2d(128,256,k3,p1) -> ReLU -> BN: w4 = conv.4.weight in (ky, kx, ci) K order [256][1152], c2 channel-last [P][7][7][128]
hipError_t launch_mask_conv2_t16(hipStream_t s, const float* w4, const float* c2, const float* bias, const float* scale,
                                 const float* shift, float* V, int P, float* slab, int tile_base) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
  GemmOperand A{w4, 1152, nullptr, 0, nullptr};
  GemmOperand B{c2, 0, nullptr, 0, nullptr};
  return launch_t16c<Tile16C<B_CONV2>, EpiConvT16>(s, A, B, 256, P * kUHW, 1152, slab, EpiConvT16{V, bias, scale, shift, 256}, tile_base);
}

// How many column tiles of a P-pair launch the fused kernel takes: the whole rounds of one workgroup per CU, when there are at
// least two of them (below that the two single-convolution launches' stream-K fills the chip better) -- and the leftover tiles
// too when they fill most of another round (a round at >= 70 % of the grid costs less than the two stream-K launches + fix-ups
// over the same tiles: 16x12 x 64 clips leaves 216 of 256, 375 us against 440); 0 = not fused
int pair_convs_fused_tiles(int P) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30)) return 0;
  const int tiles = (P * kUHW + 127) / 128;
  const int G = num_cus() * kTiles[TILE_256x128].blocks_per_cu;
  const int rounds = tiles / G, left = tiles - rounds * G;
  if (rounds < 2) return 0;
  return left * 10 >= G * 7 ? tiles : rounds * G;
}

// conv3x3 -> ReLU -> BN, then the union conv on the same accumulators, for the first `ntiles` column tiles (a multiple of the
// grid, or all of them: pair_convs_fused_tiles); the caller runs the two launches above with tile_base = ntiles for the rest
hipError_t launch_pair_convs_fused_t16(hipStream_t s, const float* w4, const float* c2, const float* bias4, const float* scale,
                                       const float* shift, const float* U, const int64_t* u_off, const float* W, const float* bias1,
                                       float* V, int P, int K, int ntiles) {
  static DeviceMarks marks;
  using T = Tile16C<B_CONV2>;
  auto kern = pair_conv_fused_kernel<0>;
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int G = num_cus() * kTiles[TILE_256x128].blocks_per_cu;
  const int tiles_all = (P * kUHW + 127) / 128;
  if (P <= 0 || K % kBK != 0 || ntiles <= 0 || ntiles > tiles_all || (ntiles % G != 0 && ntiles != tiles_all)) return hipErrorInvalidValue;
  GemmOperand A2{w4, 1152, nullptr, 0, nullptr}, B2{c2, 0, nullptr, 0, nullptr};
  GemmOperand A1{W, (int64_t)K, nullptr, 0, nullptr}, B1{U, (int64_t)K * kUHW, nullptr, P, u_off};
  hipLaunchKernelGGL(kern, dim3(G), dim3(T::NT), T::LDS_BYTES, s, A2, B2, A1, B1, P * kUHW, K, ntiles,
                     EpiConvT16{V, bias4, scale, shift, 256}, EpiUnionT16{V, bias1, 256});
  return hipGetLastError();
}

}  // namespace sttran
