// kernels_gemm.hip -- instantiations, tile planning and launchers of the fp32 MFMA GEMM.
#include "gemm_launch.h"

namespace sttran {

// (the linear GEMMs instantiate GemmTile<...> in gemm_generic)
using T256x128 = GemmTile<256, 128, 4, 2>;   // 8 waves, wave tile 64x64
using T128x128 = GemmTile<128, 128, 2, 2>;   // 4 waves, wave tile 64x64
using T128x64 = GemmTile<128, 64, 2, 2>;     // 4 waves, wave tile 64x32
using T64x64 = GemmTile<64, 64, 2, 2>;       // 4 waves, wave tile 32x32
#ifdef STTRAN_GEMM_EXPERIMENT
using TConv2 = GemmTile<256, 128, 4, 2, B_CONV2>;   // conv3x3 as implicit GEMM: all 256 output channels in one tile (B gathered once)
using TUnionFlat = GemmTile<256, 128, 4, 2, B_UNION_FLAT>;   // all 256 channels in one tile, columns = pair * 49 + hw
#endif

int num_cus() {
  static int cus[kMaxDevices] = {};
  const int dev = current_device();
  int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// below this many rows the planner keeps the 32x32x2 tiles (small-M launches are latency-bound: measured in
// tools/gemm_bench.py --shapes path16 / path16x8)
static const int64_t kT16MinRows = exp_env("STTRAN_T16_MIN_ROWS") ? atoll(exp_env("STTRAN_T16_MIN_ROWS")) : 1024;

GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int force_tile, int force_split) {
  (void)force_split;   // split-K is subsumed by the stream-K schedule
  GemmPlan best{TILE_128x128, 1};
  double best_t = 1e300;
  const int64_t ksteps = (K + kBK - 1) / kBK;
  static const int env_tile = exp_env("STTRAN_GEMM_TILE") ? atoi(exp_env("STTRAN_GEMM_TILE")) : 0;   // experiments only
  if (!force_tile && env_tile > 0 && env_tile < TILE_COUNT) force_tile = env_tile;
  if (force_tile < 0 || force_tile >= TILE_COUNT || force_tile == TILE_RETIRED_6) force_tile = 0;
  for (int t = 1; t < TILE_COUNT; ++t) {
    if (force_tile && t != force_tile) continue;
    // the 176-column tile only serves N = 11 k x 16 exactly (1936 / 3872 / 5808); gemm_linear falls back to 256 x 128
    // when the operands / epilogue do not meet its contract
    if (t == TILE_RETIRED_6) continue;
    if (t == TILE_128x176 && (N % 176 != 0 || (!force_tile && M < kT16MinRows))) continue;
    if (t == TILE_T128x128 && (N % 128 != 0 || (!force_tile && M < kT16MinRows))) continue;
    const TileInfo& ti = kTiles[t];
    const int64_t tm = (M + ti.bm - 1) / ti.bm, tn = (N + ti.bn - 1) / ti.bn, tiles = tm * tn;
    const int G = grid_of(t, tiles, ksteps);
    const int cus = (int)std::min<int64_t>(G, num_cus());
    // cycles on one CU: its equal share of the (padded) MFMA steps at the tile's efficiency ...
    const double mfma = (double)tiles * ksteps / cus * ti.bm * ti.bn * kBK * 2.0 / 256.0 / ti.eff;
    // ... plus parking and re-reading the partial tiles (~2 per workgroup) at ~4 KB/cycle chip-wide (they mostly
    // stay in L2 / the Infinity Cache), the second launch, and the per-launch fixed cost
    const SkPlan sp = sk_plan(t, tiles, ksteps);
    const bool split = sp.tiles_sk != 0;
    const double fix = split ? 2.0 * (sp.g_sk + sp.tiles_sk) * ti.bm * ti.bn * 4.0 / 4000.0 + 4000.0 : 0.0;
    const double time = mfma + fix + 6000.0;
    if (time < best_t) { best_t = time; best = {t, 1}; }
  }
  return best;
}

size_t gemm_slab_floats(const GemmPlan& p, int64_t M, int64_t N) {
  (void)M; (void)N;
  const TileInfo& ti = kTiles[p.tile];
  return (size_t)num_cus() * ti.blocks_per_cu * 2 * ti.bm * ti.bn;
}

size_t gemm_slab_floats_max() {
  static size_t cached[kMaxDevices] = {};
  const int dev = current_device();
  if (!cached[dev]) {
    size_t m = 0;
    for (int t = 1; t < TILE_COUNT; ++t) if (t != TILE_RETIRED_6) m = std::max(m, gemm_slab_floats(GemmPlan{t, 1}, 0, 0));
    cached[dev] = (m + 63) & ~size_t(63);
  }
  return cached[dev];
}
size_t gemm_slab_bytes() { return gemm_slab_floats_max() * 4; }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// can EpiLinear run as 16-byte vectors?  (every pointer it dereferences at a column that is a multiple of 4)
static bool epi_vectorizable(const EpiLinear& e, int N) {
  return (N & 3) == 0 && aligned16(e.C) && (e.ldc & 3) == 0 && aligned16(e.bias) && aligned16(e.rowbias) &&
         (e.rb_ld & 3) == 0 && (e.rb_cols & 3) == 0 && aligned16(e.scale) && aligned16(e.shift) && aligned16(e.res) &&
         (e.ldres & 3) == 0;
}

// The tile a launch really runs on: the planner's choice, unless that is a 16x16x4 tile whose contract the operands or the
// epilogue do not meet (gathered-by-offset rows, a per-column affine, a gathered operand of unknown span, ...): then the
// general 32x32x2 engine's 256 x 128 tile.  (The per-launch profile records are labelled with THIS tile.)
int gemm_effective_tile(const GemmOperand& A, const GemmOperand& B, int N, int K, const EpiLinear& epi, GemmPlan plan, int padded) {
  if (plan.tile != TILE_128x176 && plan.tile != TILE_T128x128) return plan.tile;
  // gemm16_kernel addresses by (64-bit tile base + 32-bit in-tile byte offset): any operand size, as long as one tile's
  // rows stay inside 4 GB (ld < 2^21 floats); a GATHERED A has no tile base -- its rows must be known to lie within 4 GB
  // of A.ptr (GemmOperand::span; unknown = the general engine, which carries 64-bit pointers)
  const int64_t kLim = (int64_t)1 << 32;
  const bool a_ok = A.rowidx ? (A.span > 0 && (A.span * A.ld + ((K + 31) / 32) * 32) * 4 < kLim) : (129 * A.ld * 4 < kLim);
  const bool b_ok = !B.rowidx && !B.rowoff && 177 * B.ld * 4 < kLim;
  // rows gathered by element offset (rowoff): the 128 x 128 tile has a 64-bit-pointer form (K % 32 == 0: a gathered caller
  // tensor has no zero padding behind its rows), the 176-column tile does not
  const bool ro_ok = !A.rowoff || (plan.tile == TILE_T128x128 && !A.rowidx && K % 32 == 0 && (A.aux == 0 || A.aux % 128 == 0));
  const bool ok = padded && N % (plan.tile == TILE_T128x128 ? 128 : 176) == 0 && epi_vectorizable(epi, N) && !epi.scale && ro_ok &&
                  aligned16(A.ptr) && (A.ld & 3) == 0 && aligned16(B.ptr) && (B.ld & 3) == 0 && a_ok && b_ok;
  return ok ? plan.tile : TILE_256x128;
}

// padded != 0: the operands meet the B_KMAJOR_PAD contract (gemm_f32_mfma.h); rows must then be 128-byte multiples apart
// only as far as the caller's ld says -- what matters is that ceil32(K) columns of every row are readable.
hipError_t gemm_linear(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                       const EpiLinear& epi, GemmPlan plan, float* slab, int padded) {
  if (M <= 0 || N <= 0) return hipSuccess;
  // swapped MFMA ports; 16-byte vector epilogue when every pointer allows it, else the same kernel with scalar stores
  // (arbitrary caller tensors -- the select path of sttran_debug_gemm -- always take the scalar form)
  if (plan.tile == TILE_128x176 || plan.tile == TILE_T128x128) {
    if (gemm_effective_tile(A, B, N, K, epi, plan, padded) == plan.tile) return gemm_linear_t16(s, A, B, M, N, K, epi, slab, plan.tile);
    plan.tile = TILE_256x128;                          // contract not met: the general engine
  }
  if (!padded) return gemm_linear_sel(s, A, B, M, N, K, epi, plan, slab);
  if (epi_vectorizable(epi, N)) return gemm_linear_vec(s, A, B, M, N, K, epi, plan, slab);
  return gemm_linear_s4(s, A, B, M, N, K, epi, plan, slab);
}
hipError_t gemm_heads(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                      const EpiHeads& epi, GemmPlan plan, float* slab) {
  if (M <= 0 || N <= 0) return hipSuccess;
  // N = 26: only the 64x64 tile makes sense (api_forward.hip forces it); padded operands
  return launch_tile<GemmTile<64, 64, 2, 2, B_KMAJOR_PAD>, EpiScalar4<EpiHeads>>(s, TILE_64x64, A, B, M, N, K, slab, EpiScalar4<EpiHeads>{epi});
}
#ifdef STTRAN_GEMM_EXPERIMENT   // round 2's 32x32x2 forms of the two convolutions: A/B runs only (STTRAN_CONV_ENGINE=32x32)
// union_func1: M = 256 out channels (A = W[256][K], one M-tile), N = 49 P columns (pair, hw) read in place from the
// NCHW tensor (B_UNION_FLAT), hybrid data-parallel + stream-K schedule like every other GEMM
hipError_t launch_union_conv(hipStream_t s, const float* U, const int64_t* u_off, const float* W, const float* bias, float* V,
                             int P, int K, float* slab) {
  if (K % kBK != 0 || P <= 0 || (int64_t)P * kUHW >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
  GemmOperand A{W, (int64_t)K, nullptr, 0};
  GemmOperand B{U, (int64_t)K * kUHW, nullptr, P, u_off};
  EpiUnionFlat epi{V, bias, 256, P};
  return launch_tile<TUnionFlat, EpiUnionFlat>(s, TILE_256x128, A, B, 256, P * kUHW, K, slab, epi);
}

// Conv2d(128,256,k3,p1) -> ReLU -> BN as implicit GEMM: A = conv.4.weight.view(256, 1152), B gathered
// from C2 [P,128,7,7], output V[p][c][49]
hipError_t launch_mask_conv2(hipStream_t s, const float* w4, const float* c2, const EpiConvRelBn& epi, int P,
                             float* slab) {
  GemmOperand A{w4, 1152, nullptr, 0};
  GemmOperand B{c2, 0, nullptr, 0};
  return launch_tile<TConv2, EpiConvRelBn>(s, TILE_256x128, A, B, 256, P * 49, 1152, slab, epi);
}

#endif

// ---- calibration: back-to-back v_mfma_f32_32x32x2_f32 on independent accumulators, no memory ----
// Gives the fp32-MFMA rate THIS device sustains (clock under load differs between MI355X boards by
// several per cent), the yardstick tools/gemm_bench.py normalises against.
__global__ void __launch_bounds__(256) mfma_peak_kernel(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 1.f; a2[e] = 2.f; a3[e] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  if (s == 12345.678f) out[0] = s;     // keep the chain alive
}

hipError_t launch_mfma_peak(hipStream_t s, float* out, int iters, int blocks) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, out, iters);
  return hipGetLastError();
}

}  // namespace sttran
