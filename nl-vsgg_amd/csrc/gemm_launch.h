// gemm_launch.h -- tile table, stream-K schedule and launch templates of the fp32 MFMA GEMM, shared by the translation
// units that instantiate it (one per epilogue family, so that `make -j` compiles them side by side).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kernels.h"

namespace sttran {

struct TileInfo { int bm, bn; float eff; int blocks_per_cu; };
// eff = fraction of the fp32-MFMA peak the tile's main loop sustains on a large square problem
// (tools/gemm_bench.py --shapes big on MI355X); blocks_per_cu = persistent workgroups per CU
// (bounded by LDS: 111 / 74 / 37 / 55 KB per workgroup).
static const TileInfo kTiles[TILE_COUNT] = {
    {0, 0, 0.f, 0}, {256, 128, 0.90f, 1}, {128, 128, 0.88f, 2}, {64, 64, 0.79f, 4}, {128, 64, 0.85f, 2},
    {128, 176, 0.93f, 2},    // gemm_f32_t16.h: 76 KB of LDS, two workgroups of 4 waves per CU
    {0, 0, 0.f, 0},          // (id 6 retired: the 256 x 176 form)
    {128, 128, 0.92f, 2}};   // ... 64 KB of LDS, N % 128 == 0 (2048, 512)

// Hybrid data-parallel + stream-K schedule of one GEMM: G persistent workgroups each run dp_per_wg whole
// tiles; the tiles_sk leftover tiles (< G) are cut into g_sk equal iteration ranges.
struct SkPlan { int G, dp_per_wg, tiles_sk, g_sk; };
static SkPlan sk_plan(int tile, int64_t tiles, int64_t ksteps) {
  int64_t g = (int64_t)num_cus() * kTiles[tile].blocks_per_cu;
#ifdef STTRAN_GEMM_EXPERIMENT
  if (tile == TILE_128x176 && getenv("STTRAN_T16_BPC")) g = (int64_t)num_cus() * atoi(getenv("STTRAN_T16_BPC"));
#endif
  SkPlan p;
  // never cut finer than kMinSteps K-steps per workgroup: below that the per-segment prologue dominates
  static const int kMinSteps = exp_env("STTRAN_SK_MIN_STEPS") ? std::max(1, atoi(exp_env("STTRAN_SK_MIN_STEPS"))) : 4;
  p.G = (int)std::min<int64_t>(g, std::max<int64_t>(1, tiles * ksteps / kMinSteps));
  p.dp_per_wg = (int)(tiles / p.G);
  p.tiles_sk = (int)(tiles - (int64_t)p.dp_per_wg * p.G);
  p.g_sk = (int)std::max<int64_t>(p.tiles_sk ? 1 : 0, std::min<int64_t>(p.G, (int64_t)p.tiles_sk * ksteps / kMinSteps));
  return p;
}
static int grid_of(int tile, int64_t tiles, int64_t ksteps) { return sk_plan(tile, tiles, ksteps).G; }

#ifndef STTRAN_GEMM_PIPE
#define STTRAN_GEMM_PIPE 1
#endif

template <class T, class Epi, int PIPE>
static hipError_t launch_tile_p(hipStream_t s, int tile_id, const GemmOperand& A, const GemmOperand& B, int M, int N,
                                int K, float* slab, const Epi& epi) {
  static DeviceMarks marks;
  auto kern = gemm_sk_kernel<T, Epi, PIPE>;
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int tm = (M + T::BM - 1) / T::BM, tn = (N + T::BN - 1) / T::BN, tiles = tm * tn;
  const int ksteps = (K + kBK - 1) / kBK;
  const SkPlan sp = sk_plan(tile_id, tiles, ksteps);
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  // more than one workgroup per CU: the second-dispatched ones walk their work in the opposite order (see the kernel)
  static const int env_stagger = exp_env("STTRAN_GEMM_STAGGER") ? atoi(exp_env("STTRAN_GEMM_STAGGER")) : 1;
  const int half = (env_stagger && sp.G > num_cus()) ? std::max(num_cus(), sp.G / 2) : sp.G;
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, tm, tiles, ksteps, sp.dp_per_wg,
                     sp.g_sk, base, rem, half, slab, epi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !split) return e;
  if constexpr (EpiTraits<Epi>::swap)
    hipLaunchKernelGGL((gemm_fixup_vec_kernel<T, Epi>), dim3(sp.tiles_sk, T::TM * T::TN * 4), dim3(T::NT), 0, s, M, N, tm, tn,
                       ksteps, sp.g_sk, base, rem, tiles - sp.tiles_sk, slab, epi);
  else
    hipLaunchKernelGGL((gemm_fixup_kernel<T, Epi>), dim3(sp.tiles_sk, T::TM * T::TN * 4), dim3(T::NT), 0, s, M, N, tm, tn,
                       ksteps, sp.g_sk, base, rem, tiles - sp.tiles_sk, slab, epi);
  return hipGetLastError();
}

template <class T, class Epi>
static hipError_t launch_tile(hipStream_t s, int tile_id, const GemmOperand& A, const GemmOperand& B, int M, int N,
                              int K, float* slab, const Epi& epi) {
#ifdef STTRAN_GEMM_EXPERIMENT
  // build-time experiment switch (tools/gemm_bench.py): pick the main-loop variant at run time
  const char* v = getenv("STTRAN_GEMM_PIPE");
  const int pipe = v ? atoi(v) : STTRAN_GEMM_PIPE;
  if (pipe == 1) return launch_tile_p<T, Epi, 1>(s, tile_id, A, B, M, N, K, slab, epi);
  if (pipe == 2) return launch_tile_p<T, Epi, 2>(s, tile_id, A, B, M, N, K, slab, epi);
  if (pipe == 3) return launch_tile_p<T, Epi, 3>(s, tile_id, A, B, M, N, K, slab, epi);
  if (pipe == 4) return launch_tile_p<T, Epi, 4>(s, tile_id, A, B, M, N, K, slab, epi);
  if (pipe == 5) return launch_tile_p<T, Epi, 5>(s, tile_id, A, B, M, N, K, slab, epi);
  return launch_tile_p<T, Epi, 0>(s, tile_id, A, B, M, N, K, slab, epi);
#else
  return launch_tile_p<T, Epi, STTRAN_GEMM_PIPE>(s, tile_id, A, B, M, N, K, slab, epi);
#endif
}

template <class Epi, int BK>
static hipError_t gemm_generic(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                               const Epi& epi, GemmPlan plan, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  switch (plan.tile) {
    case TILE_256x128: return launch_tile<GemmTile<256, 128, 4, 2, BK>, Epi>(s, plan.tile, A, B, M, N, K, slab, epi);
    case TILE_128x128: return launch_tile<GemmTile<128, 128, 2, 2, BK>, Epi>(s, plan.tile, A, B, M, N, K, slab, epi);
    case TILE_128x64: return launch_tile<GemmTile<128, 64, 2, 2, BK>, Epi>(s, plan.tile, A, B, M, N, K, slab, epi);
    default: return launch_tile<GemmTile<64, 64, 2, 2, BK>, Epi>(s, TILE_64x64, A, B, M, N, K, slab, epi);
  }
}


// the three instantiation families of nn.Linear GEMMs (kernels_gemm_vec.hip / _s4.hip / _sel.hip)
hipError_t gemm_linear_vec(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, GemmPlan plan, float* slab);      // padded operands, 16-byte epilogue
hipError_t gemm_linear_s4(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                          const EpiLinear& epi, GemmPlan plan, float* slab);       // padded operands, scalar epilogue
hipError_t gemm_linear_sel(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, GemmPlan plan, float* slab);      // arbitrary operands (zero-select)
// the 128 x 176 tile on 16x16x4 MFMA blocks (kernels_gemm_t16.hip): N % 176 == 0, padded operands, 16-byte epilogue
hipError_t gemm_linear_t16(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, float* slab, int tile);

}  // namespace sttran
