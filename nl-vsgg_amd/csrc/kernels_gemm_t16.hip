// kernels_gemm_t16.hip -- the 128 x 176 tile of the N = 1936 family (gemm_f32_t16.h): instantiation and launch
#include <cstdio>

#include "gemm_f32_t16.h"
#include "gemm_launch.h"

namespace sttran {

template <class T, int TILE_ID, bool ROWOFF = false>
static hipError_t launch_t16(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                             const EpiLinear& epi, float* slab) {
  using Epi = EpiLinearV;
  if (M <= 0 || N <= 0) return hipSuccess;
  if (N % T::BN != 0 || (ROWOFF != (A.rowoff != nullptr))) return hipErrorInvalidValue;
  static DeviceMarks marks;
  auto kern = gemm16_kernel<T, Epi, 0, ROWOFF>;
#ifdef STTRAN_GEMM_EXPERIMENT
  {
    static DeviceMarks m1, m2, m3, m4, m5, m9;
    const char* v = getenv("STTRAN_T16_ABLATE");
    const int abl = v ? atoi(v) : 0;
    if (abl == 1) { kern = gemm16_kernel<T, Epi, 1>; if (m1.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
    if (abl == 2) { kern = gemm16_kernel<T, Epi, 2>; if (m2.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
    if (abl == 3) { kern = gemm16_kernel<T, Epi, 3>; if (m3.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
    if (abl == 5) { kern = gemm16_kernel<T, Epi, 5>; if (m5.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
    if (abl == 9) { kern = gemm16_kernel<T, Epi, 9>; if (m9.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
    if (abl == 4) { kern = gemm16_kernel<T, Epi, 4>; if (m4.raise_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES) != hipSuccess) return hipErrorUnknown; }
  }
#endif
  {
    hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(gemm16_kernel<T, Epi, 0, ROWOFF>), T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
#ifdef STTRAN_GEMM_EXPERIMENT
  {
    static int skew_set = -1;
    const int skew = getenv("STTRAN_T16_SKEW") ? atoi(getenv("STTRAN_T16_SKEW")) : 0;
    if (skew != skew_set) {
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_t16_skew), &skew, sizeof(int)) != hipSuccess) return hipErrorUnknown;
      skew_set = skew;
    }
  }
#endif
#ifdef STTRAN_GEMM_EXPERIMENT
  {
    static bool once = false;
    if (!once && getenv("STTRAN_T16_OCC")) {
      once = true;
      int nb = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm16_kernel<T, Epi>, T::NT, T::LDS_BYTES);
      hipFuncAttributes fa{};
      hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(gemm16_kernel<T, Epi>));
      fprintf(stderr, "t16 occupancy: %d blocks/CU (err %d), lds dyn %d static %zu, regs %d, maxDyn %d\n", nb, (int)e, T::LDS_BYTES,
              fa.sharedSizeBytes, fa.numRegs, fa.maxDynamicSharedSizeBytes);
    }
  }
#endif
  const int tm = (M + T::BM - 1) / T::BM, tn = N / T::BN, tiles = tm * tn;
  const int ksteps = (K + kBK - 1) / kBK;
  if constexpr (T::BN == 128 && !ROWOFF) {
    // A narrow, deep launch ([11264, 512, 12544] = vr_fc: 4 N-tiles, 392 K-steps): gangs of `tn` workgroups walk one
    // stream-K range over (M-panel, K-step) in lockstep, so an A panel is read once, not once per N-tile (gemm16_kernel GANG)
    const int Gw = num_cus() * kTiles[TILE_ID].blocks_per_cu;
    static const int env_gang = exp_env("STTRAN_T16_GANG") ? atoi(exp_env("STTRAN_T16_GANG")) : 1;
    // GM = 2 M-panels per gang when that divides the grid: the members on one N-tile then share their weight panel too
    const int gm = (env_gang >= 2 ? env_gang - 1 : 2);
    const int GMv = (tm >= 2 && Gw % (tn * gm) == 0) ? gm : 1;
    if (env_gang && tn >= 2 && tn <= 8 && Gw % (tn * GMv) == 0 && ksteps >= 128 && tiles < 2 * Gw &&
        (int64_t)((tm + GMv - 1) / GMv) * ksteps >= (int64_t)(Gw / (tn * GMv)) * 8) {
#ifdef STTRAN_GEMM_EXPERIMENT
      if (getenv("STTRAN_T16_ABLATE") && atoi(getenv("STTRAN_T16_ABLATE")) != 0) goto plain;
#endif
      {
        static DeviceMarks gmarks;
        auto gk = gemm16_kernel<T, Epi, 0, false, true>;
        hipError_t e0 = gmarks.raise_lds(reinterpret_cast<const void*>(gk), T::LDS_BYTES);
        if (e0 != hipSuccess) return e0;
        const int gangs = Gw / (tn * GMv);
        const int64_t total = (int64_t)((tm + GMv - 1) / GMv) * ksteps;      // (group of GM panels, K-step)
        if (total >= (int64_t)1 << 30 || !slab) return hipErrorInvalidValue;
        const int base = (int)(total / gangs), rem = (int)(total % gangs);
        const Epi e{epi};
        hipLaunchKernelGGL(gk, dim3(Gw), dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, tm, tiles, ksteps, GMv, gangs, base, rem, tn, slab, e);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL((gemm16_fixup_kernel<T, Epi, true>), dim3(tm * tn, 2 * T::NB), dim3(T::NT), 0, s, M, N, tm, tn, ksteps, gangs,
                           base, rem, GMv, tn, slab, e);
        return hipGetLastError();
      }
    }
  }
#ifdef STTRAN_GEMM_EXPERIMENT
plain:
#endif
  const SkPlan sp = sk_plan(TILE_ID, tiles, ksteps);
  const int64_t total = (int64_t)sp.tiles_sk * ksteps;
  if (total >= (int64_t)1 << 30) return hipErrorInvalidValue;
  const int base = sp.g_sk ? (int)(total / sp.g_sk) : 0, rem = sp.g_sk ? (int)(total % sp.g_sk) : 0;
  bool split = false;
  for (int b = 1; b < sp.g_sk && !split; ++b) split = (sk_range(b, base, rem).begin % ksteps) != 0;
  if (split && !slab) return hipErrorInvalidValue;
  // N-tiles per group of the tile order (consecutive tile indices sweep `half` N-tiles of one M-tile, then the next M-tile):
  // the 64 workgroups of an XCD hold a (64 / half) x half block of tiles whose panels they share through its L2
  static const int env_gn = exp_env("STTRAN_T16_GROUP_N") ? atoi(exp_env("STTRAN_T16_GROUP_N")) : 0;
  const int half = env_gn > 0 ? env_gn : T::GROUP_N;
  const Epi e{epi};
  hipLaunchKernelGGL(kern, dim3(sp.G), dim3(T::NT), T::LDS_BYTES, s, A, B, M, N, K, tm, tiles, ksteps, sp.dp_per_wg, sp.g_sk,
                     base, rem, half, slab, e);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess || !split) return err;
  hipLaunchKernelGGL((gemm16_fixup_kernel<T, Epi>), dim3(sp.tiles_sk, 2 * T::NB), dim3(T::NT), 0, s, M, N, tm, tn, ksteps,
                     sp.g_sk, base, rem, tiles - sp.tiles_sk, half, slab, e);
  return hipGetLastError();
}

hipError_t gemm_linear_t16(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                           const EpiLinear& epi, float* slab, int tile) {
  if (tile == TILE_T128x128 && A.rowoff) return launch_t16<Tile16<128, 128>, TILE_T128x128, true>(s, A, B, M, N, K, epi, slab);
  if (tile == TILE_T128x128) return launch_t16<Tile16<128, 128>, TILE_T128x128>(s, A, B, M, N, K, epi, slab);
  return launch_t16<Tile16<128, 176>, TILE_128x176>(s, A, B, M, N, K, epi, slab);
}

#ifdef STTRAN_GEMM_EXPERIMENT
// trace buffer of the ABL == 9 build (tools/experiments/t16_trace.py): buf = device memory for cap records of 8 uint64, or NULL
extern "C" int sttran_debug_t16_trace(unsigned long long* buf, unsigned int cap) {
  const unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_t16_trace), &buf, sizeof(buf)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_t16_trace_cap), &cap, sizeof(cap)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_t16_trace_n), &zero, sizeof(zero)) != hipSuccess)
    return 2;
  return 0;
}
extern "C" int sttran_debug_t16_trace_count(unsigned int* n) {
  return hipMemcpyFromSymbol(n, HIP_SYMBOL(g_t16_trace_n), sizeof(unsigned int)) == hipSuccess ? 0 : 2;
}
#endif
#ifdef STTRAN_GEMM_EXPERIMENT
// reads and clears the phase clocks (tools/gemm_bench.py --phases)
extern "C" int sttran_debug_t16_clocks(unsigned long long* out8) {
  unsigned long long z[8] = {};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_t16_clk), sizeof(z)) != hipSuccess) return 2;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_t16_clk), z, sizeof(z)) == hipSuccess ? 0 : 2;
}
#endif

}  // namespace sttran
