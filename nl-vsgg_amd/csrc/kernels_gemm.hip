// kernels_gemm.hip -- instantiations, tile planning and launchers of the fp32 MFMA GEMM.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kernels.h"

namespace sttran {

using T256x256 = GemmTile<256, 256, 4, 2>;   // 8 waves, wave tile 64x128 (2x4 MFMA tiles)
using T256x128 = GemmTile<256, 128, 4, 2>;   // 8 waves, wave tile 64x64
using T128x128 = GemmTile<128, 128, 2, 2>;   // 4 waves, wave tile 64x64
using T128x64 = GemmTile<128, 64, 2, 2>;     // 4 waves, wave tile 64x32
using T64x64 = GemmTile<64, 64, 2, 2>;       // 4 waves, wave tile 32x32

struct TileInfo { int bm, bn; float eff; int blocks_per_cu; };
// eff = measured fraction of the fp32-MFMA peak a full grid of that tile sustains (see
// DESIGN.md "GEMM tile planning"); used only to rank plans.
static const TileInfo kTiles[TILE_COUNT] = {
    {0, 0, 0.f, 0}, {256, 256, 0.80f, 1}, {128, 128, 0.72f, 2}, {64, 64, 0.50f, 4},
    {256, 128, 0.76f, 1}, {128, 64, 0.62f, 3}};

constexpr int kNumCU = 256;

GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int force_tile, int force_split) {
  GemmPlan best{TILE_128x128, 1};
  double best_t = 1e300;
  for (int t = 1; t < TILE_COUNT; ++t) {
    if (force_tile && t != force_tile) continue;
    const TileInfo& ti = kTiles[t];
    const int64_t tm = (M + ti.bm - 1) / ti.bm, tn = (N + ti.bn - 1) / ti.bn;
    for (int sk = 1; sk <= 16; sk *= 2) {
      if (force_split && sk != force_split) continue;
      int64_t kchunk = ((K + sk - 1) / sk + kBK - 1) / kBK * kBK;
      if (sk > 1 && kchunk * (sk - 1) >= K) continue;     // an empty split
      if (sk > 1 && kchunk < 256) continue;
      const int64_t blocks = tm * tn * sk;
      const int64_t per_cu = (blocks + kNumCU - 1) / kNumCU;     // critical-path blocks on one CU
      // MFMA time of one block (cycles on its CU) / efficiency
      double t_blk = (double)ti.bm * ti.bn * kchunk * 2.0 / 256.0 / ti.eff;
      double time = per_cu * t_blk + 6000.0;                      // + launch/prologue
      if (sk > 1) time += 4000.0 + (double)M * N * (sk + 1) * 4.0 / (2000.0);  // slab write+reduce
      if (time < best_t) { best_t = time; best = {t, sk}; }
    }
  }
  return best;
}

size_t gemm_slab_floats(const GemmPlan& p, int64_t M, int64_t N) {
  return p.splitk > 1 ? (size_t)p.splitk * M * N : 0;
}

template <class T, class Epi>
static hipError_t launch_tile(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                              int splitk, const Epi& epi) {
  static bool attr_set = false;
  auto kern = gemm_nt_kernel<T, Epi>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tm = (M + T::BM - 1) / T::BM, tn = (N + T::BN - 1) / T::BN;
  const int kchunk = ((K + splitk - 1) / splitk + kBK - 1) / kBK * kBK;
  dim3 grid(tm * tn, 1, splitk);
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, kchunk, tm, epi);
  return hipGetLastError();
}

template <class Epi>
static hipError_t launch_any(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                             int tile, int splitk, const Epi& epi) {
  switch (tile) {
    case TILE_256x256: return launch_tile<T256x256, Epi>(s, A, B, M, N, K, splitk, epi);
    case TILE_256x128: return launch_tile<T256x128, Epi>(s, A, B, M, N, K, splitk, epi);
    case TILE_128x128: return launch_tile<T128x128, Epi>(s, A, B, M, N, K, splitk, epi);
    case TILE_128x64: return launch_tile<T128x64, Epi>(s, A, B, M, N, K, splitk, epi);
    default: return launch_tile<T64x64, Epi>(s, A, B, M, N, K, splitk, epi);
  }
}

template <class Epi>
static hipError_t gemm_generic(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                               const Epi& epi, GemmPlan plan, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if (plan.splitk <= 1) return launch_any<Epi>(s, A, B, M, N, K, plan.tile, 1, epi);
  if (!slab) return hipErrorInvalidValue;
  EpiSlab es{slab, (int64_t)N, (int64_t)M * N};
  hipError_t e = launch_any<EpiSlab>(s, A, B, M, N, K, plan.tile, plan.splitk, es);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)M * N;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL((splitk_reduce_kernel<Epi>), dim3(blocks), dim3(256), 0, s, slab, plan.splitk, M, N,
                     (int64_t)M * N, epi);
  return hipGetLastError();
}

hipError_t gemm_linear(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                       const EpiLinear& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiLinear>(s, A, B, M, N, K, epi, plan, slab);
}
hipError_t gemm_heads(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                      const EpiHeads& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiHeads>(s, A, B, M, N, K, epi, plan, slab);
}
hipError_t gemm_conv(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                     const EpiConvRelBn& epi, GemmPlan plan, float* slab) {
  return gemm_generic<EpiConvRelBn>(s, A, B, M, N, K, epi, plan, slab);
}

}  // namespace sttran
