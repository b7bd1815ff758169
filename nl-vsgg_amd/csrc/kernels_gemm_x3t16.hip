// kernels_gemm_x3t16.hip -- the bf16x3 engine on the 128 x 176 / 128 x 128 tiles (gemm_bf16x3_t16.h): instantiation and launch
#include "gemm_bf16x3_t16c.h"
#include "gemm_launch.h"

namespace sttran {

size_t fm_planes_bytes(int64_t rows, int64_t K) {
  return (size_t)((rows + 15) / 16) * (size_t)((K + 31) / 32) * kFmBlock3 * sizeof(__bf16);
}

hipError_t split_fm(hipStream_t s, const float* src, int64_t ld, const int32_t* rowidx, const int64_t* rowoff, int M, int K,
                    void* planes, int weight) {
  if (M <= 0 || K <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (ld & 3) || (reinterpret_cast<uintptr_t>(planes) & 15)) return hipErrorInvalidValue;
  const int kb = (K + 31) / 32;
  const int64_t blocks = (int64_t)((M + 15) / 16) * kb;
  hipLaunchKernelGGL(split_fm_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, s, src, ld, rowidx, rowoff, M, K, kb, blocks,
                     reinterpret_cast<__bf16*>(planes), weight ? 0 : 1);
  return hipGetLastError();
}

template <class T, int TILE_ID, class Epi>
static hipError_t launch_x3t16(hipStream_t s, const FmPlanes& A, const FmPlanes& B, int M, int N, int K, const Epi& e,
                               float* slab) {
  static DeviceMarks marks;
  auto kern = gemm16x3_kernel<T, Epi>;
#ifdef STTRAN_GEMM_EXPERIMENT
  {
    static DeviceMarks m[7];
    const int abl = getenv("STTRAN_X3_ABLATE") ? atoi(getenv("STTRAN_X3_ABLATE")) : 0;
    if (abl == 1) kern = gemm16x3_kernel<T, Epi, 1>;
    if (abl == 2) kern = gemm16x3_kernel<T, Epi, 2>;
    if (abl == 3) kern = gemm16x3_kernel<T, Epi, 3>;
    if (abl == 4) kern = gemm16x3_kernel<T, Epi, 4>;
    if (abl == 5) kern = gemm16x3_kernel<T, Epi, 5>;
    if (abl == 6) kern = gemm16x3_kernel<T, Epi, 6>;
    if (abl >= 1 && abl <= 6 && m[abl].raise_lds(reinterpret_cast<const void*>(kern), X3T16<T>::LDS_BYTES) != hipSuccess) return hipErrorUnknown;
  }
#endif
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(gemm16x3_kernel<T, Epi>), X3T16<T>::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int tm = (M + T::BM - 1) / T::BM, tn = N / T::BN, tiles = tm * tn;
  const int ksteps = (K + kBK - 1) / kBK;
  const SkPlan sp = sk_plan(TILE_ID, tiles, ksteps);
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  // N-tiles per group of the tile order (see launch_t16); experiment builds: STTRAN_X3_GROUP_N
  static const int env_gn = exp_env("STTRAN_X3_GROUP_N") ? atoi(exp_env("STTRAN_X3_GROUP_N")) : 0;
  const int half = env_gn > 0 ? env_gn : T::GROUP_N;
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), X3T16<T>::LDS_BYTES, s, A, B, M, N, K, tm, tiles, ksteps, sp.dp_per_wg, sp.g_sk,
                     base, rem, half, slab, e);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess || !split) return err;
  hipLaunchKernelGGL((gemm16_fixup_kernel<T, Epi>), dim3(sp.tiles_sk, 2 * T::NB), dim3(T::NT), 0, s, M, N, tm, tn, ksteps, sp.g_sk,
                     base, rem, tiles - sp.tiles_sk, half, slab, e);
  return hipGetLastError();
}

static bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the tile this engine would run [M,N,K] on: TILE_128x176, TILE_T128x128, or 0 = not served (N is no multiple of 176 or 128,
// or the epilogue's operands do not allow 16-byte accesses: the caller falls back)
int x3t16_tile(int N, const EpiLinear& epi) {
  const bool vec = al16p(epi.C) && (epi.ldc & 3) == 0 && al16p(epi.bias) && al16p(epi.rowbias) && (epi.rb_ld & 3) == 0 &&
                   (epi.rb_cols & 3) == 0 && al16p(epi.scale) && al16p(epi.shift) && al16p(epi.res) && (epi.ldres & 3) == 0;
  if (!vec) return 0;
  if (N % 176 == 0) return TILE_128x176;
  if (N % 128 == 0) return TILE_T128x128;
  return 0;
}

// a_planes: fragment-major planes of A [M, K] (split_fm); b_planes: of the weight, at its first needed ROW BLOCK (16 rows each)
hipError_t gemm_linear_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                             const EpiLinear& epi, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int tile = x3t16_tile(N, epi);
  if (!tile || !al16p(a_planes) || !al16p(b_planes) || b_row_blocks * 16 < N) return hipErrorInvalidValue;
  const int kb = (K + 31) / 32;
  const FmPlanes A{reinterpret_cast<const __bf16*>(a_planes), kb, (M + 15) / 16};
  const FmPlanes B{reinterpret_cast<const __bf16*>(b_planes), kb, b_row_blocks};
  if (tile == TILE_128x176) return launch_x3t16<Tile16<128, 176>, TILE_128x176>(s, A, B, M, N, K, EpiLinearV{epi}, slab);
  return launch_x3t16<Tile16<128, 128>, TILE_T128x128>(s, A, B, M, N, K, EpiLinearV{epi}, slab);
}

// out_planes[M, N] (fragment-major, as the NEXT launch's activation operand) = act(A W^T + bias): linear1 -> ReLU of the FFN
hipError_t gemm_act_planes_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                                 const float* bias, int relu, void* out_planes, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if (!bias || !al16p(bias) || !al16p(a_planes) || !al16p(b_planes) || !al16p(out_planes) || b_row_blocks * 16 < N || (N & 31))
    return hipErrorInvalidValue;
  const int kb = (K + 31) / 32;
  const FmPlanes A{reinterpret_cast<const __bf16*>(a_planes), kb, (M + 15) / 16};
  const FmPlanes B{reinterpret_cast<const __bf16*>(b_planes), kb, b_row_blocks};
  const EpiActPlanes e{bias, reinterpret_cast<__bf16*>(out_planes), N / 32, relu};
  if (N % 176 == 0) return launch_x3t16<Tile16<128, 176>, TILE_128x176>(s, A, B, M, N, K, e, slab);
  if (N % 128 == 0) return launch_x3t16<Tile16<128, 128>, TILE_T128x128>(s, A, B, M, N, K, e, slab);
  return hipErrorInvalidValue;
}

// C = A W^T + bias + res as fp32 rows AND as the next launch's activation planes (linear2 of a decoder layer that is not the last)
hipError_t gemm_res_planes_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                                 const EpiLinear& epi, void* out_planes, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if (!epi.bias || !epi.res || epi.relu || epi.rowbias || epi.out_rowidx || epi.out_rowidx2 || epi.scale || epi.res_rowidx ||
      !x3t16_tile(N, epi) || !al16p(out_planes) || b_row_blocks * 16 < N)
    return hipErrorInvalidValue;
  const int kb = (K + 31) / 32;
  const FmPlanes A{reinterpret_cast<const __bf16*>(a_planes), kb, (M + 15) / 16};
  const FmPlanes B{reinterpret_cast<const __bf16*>(b_planes), kb, b_row_blocks};
  const EpiResPlanes e{epi.C, epi.ldc, epi.bias, epi.res, epi.ldres, reinterpret_cast<__bf16*>(out_planes), (N + 31) / 32};
  if (N % 176 == 0) return launch_x3t16<Tile16<128, 176>, TILE_128x176>(s, A, B, M, N, K, e, slab);
  return launch_x3t16<Tile16<128, 128>, TILE_T128x128>(s, A, B, M, N, K, e, slab);
}

// ---- the two convolutions (gemm_bf16x3_t16c.h): activations loaded as fp32 and split in registers, weights fragment-major ----
template <int AKIND, class Epi>
static hipError_t launch_x3t16c(hipStream_t s, const GemmOperand& A, const FmPlanes& B, int M, int N, int K, const Epi& e, float* slab,
                                int tile_base) {
  using T = Tile16<128, 128>;
  static DeviceMarks marks;
  auto kern = gemm16x3c_kernel<T, AKIND, Epi>;
  {
    hipError_t er = marks.raise_lds(reinterpret_cast<const void*>(kern), X3T16<T>::LDS_BYTES);
    if (er != hipSuccess) return er;
  }
  const int tm = (M + T::BM - 1) / T::BM, tn = N / T::BN, tiles_all = tm * tn;
  if (tile_base < 0 || tile_base >= tiles_all || (tile_base & 1)) return hipErrorInvalidValue;
  const int tiles = tiles_all - tile_base;         // tile_base > 0: the tiles behind the fused launch's
  const int ksteps = K / kBK;
  const SkPlan sp = sk_plan(TILE_T128x128, tiles, ksteps);
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  const int half = 2;                              // both N-tiles of an M-panel side by side: the activation rows are read once
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), X3T16<T>::LDS_BYTES, s, A, B, M, N, K, tm, tiles_all, ksteps, sp.dp_per_wg, sp.g_sk,
                     base, rem, half, tile_base, slab, e);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess || !split) return err;
  hipLaunchKernelGGL((gemm16_fixup_kernel<T, Epi>), dim3(sp.tiles_sk, 2 * T::NB), dim3(T::NT), 0, s, M, N, tm, tn, ksteps, sp.g_sk,
                     base, rem, tile_base + tiles - sp.tiles_sk, half, slab, e);
  return hipGetLastError();
}

// union_func1: V[p][c][hw] += W[c][:] . U[p][:][hw] + b[c]; planes_fm = fragment-major planes of the [256, K] weight
hipError_t launch_union_conv_x3t16(hipStream_t s, const float* U, const int64_t* u_off, const void* planes_fm, const float* bias,
                                   float* V, int P, int K, float* slab, int tile_base) {
  if (K % kBK != 0 || P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30) || !al16p(planes_fm)) return hipErrorInvalidValue;
  GemmOperand A{U, (int64_t)K * kUHW, nullptr, P, u_off};
  const FmPlanes B{reinterpret_cast<const __bf16*>(planes_fm), K / 32, 16};
  return launch_x3t16c<AC_UNION, EpiUnionRows>(s, A, B, P * kUHW, 256, K, EpiUnionRows{V, bias, 256}, slab, tile_base);
}

// Conv2d(128, 256, 3, padding 1) -> ReLU -> BN: planes_fm = planes of the (ky, kx, ci)-ordered [256, 1152] weight, c2 = channel-last
// [P][7][7][128]
hipError_t launch_mask_conv2_x3t16(hipStream_t s, const void* planes_fm, const float* c2, const float* bias, const float* scale,
                                   const float* shift, float* V, int P, float* slab, int tile_base) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30) || !al16p(planes_fm) || !al16p(c2)) return hipErrorInvalidValue;
  GemmOperand A{c2, 0, nullptr, 0};
  const FmPlanes B{reinterpret_cast<const __bf16*>(planes_fm), 1152 / 32, 16};
  return launch_x3t16c<AC_CONV2, EpiConvRows>(s, A, B, P * kUHW, 256, 1152, EpiConvRows{V, bias, scale, shift, 256}, slab, tile_base);
}

// the second engine's fused pair-conv launch (gemm_bf16x3_t16c.h pair_conv_fused_x3_kernel): the first
// pair_convs_fused_tiles_x3(P) tiles (128 rows x 128 channels; both channel halves of a row panel are neighbours) of a launch =
// its whole rounds of two workgroups per CU, when there are at least two; 0 = not fused
int pair_convs_fused_tiles_x3(int P) {
  if (P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30)) return 0;
  const int tiles = ((P * kUHW + 127) / 128) * 2;
  const int G = num_cus() * kTiles[TILE_T128x128].blocks_per_cu;
  const int rounds = tiles / G, left = tiles - rounds * G;
  if (rounds < 2) return 0;
  return left * 10 >= G * 7 ? tiles : rounds * G;      // (the leftover too when it fills most of another round)
}
hipError_t launch_pair_convs_fused_x3t16(hipStream_t s, const void* w4_planes_fm, const float* c2, const float* bias4, const float* scale,
                                         const float* shift, const float* U, const int64_t* u_off, const void* wu_planes_fm,
                                         const float* bias1, float* V, int P, int K, int ntiles) {
  using T = Tile16<128, 128>;
  static DeviceMarks marks;
  auto kern = pair_conv_fused_x3_kernel<T>;
  {
    hipError_t er = marks.raise_lds(reinterpret_cast<const void*>(kern), X3T16<T>::LDS_BYTES);
    if (er != hipSuccess) return er;
  }
  const int G = num_cus() * kTiles[TILE_T128x128].blocks_per_cu;
  const int M = P * kUHW, tm = (M + T::BM - 1) / T::BM, tiles = tm * 2;
  if (P <= 0 || K % kBK != 0 || ntiles <= 0 || ntiles > tiles || (ntiles % G != 0 && ntiles != tiles) || !al16p(w4_planes_fm) || !al16p(wu_planes_fm) || !al16p(c2))
    return hipErrorInvalidValue;
  GemmOperand A2{c2, 0, nullptr, 0}, A1{U, (int64_t)K * kUHW, nullptr, P, u_off};
  const FmPlanes B2{reinterpret_cast<const __bf16*>(w4_planes_fm), 1152 / 32, 16}, B1{reinterpret_cast<const __bf16*>(wu_planes_fm), K / 32, 16};
  hipLaunchKernelGGL(kern, dim3(G), dim3(T::NT), X3T16<T>::LDS_BYTES, s, A2, B2, A1, B1, M, K, tm, tiles, ntiles, 2,
                     EpiConvRows{V, bias4, scale, shift, 256}, EpiUnionRows{V, bias1, 256});
  return hipGetLastError();
}

}  // namespace sttran
