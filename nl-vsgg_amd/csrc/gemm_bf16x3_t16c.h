// gemm_bf16x3_t16c.h -- the two convolutions of the pair fusion on the second-generation bf16x3 engine (gemm_bf16x3_t16.h):
//   union_func1  Conv2d(2048, 256, 1)       (lib/sttran.py:336,386)   rows = (pair, hw), K = 2048 input channels
//   conv.4       Conv2d(128, 256, 3, p 1)   (lib/sttran.py:342)       rows = (pair, position), K = 9 taps x 128 channels
// Same kernel skeleton: 128 x 128 tiles of v_mfma_f32_16x16x32_bf16, a wave owns 32 rows x all 128 columns, the weight
// tile's K-step arrives in LDS by LDS-DMA (fragment-major planes made once), two 4-wave workgroups per CU, gemm16_kernel's
// accumulator layout / stream-K schedule / parking format / fix-up launch.
// What differs is the ACTIVATION side: it cannot arrive pre-split -- union_feat is the model's 4.5 GB input (a split pass
// over it would cost more than the convolution), the pooled mask map is read through a 3x3 gather -- so a lane loads the
// fp32 values of its fragment itself and splits them in registers:
//   A_UNION  lane (row r = one (pair, hw), k chunk c) needs 8 consecutive channels = 8 dwords 49 floats apart (NCHW); for one
//            channel the 16 lanes of a chunk read 16 consecutive hw = 64 contiguous bytes;
//   A_CONV2  lane (row = one (pair, oy, ox), k chunk c) needs 8 consecutive channels of the channel-last map at position
//            (oy + ky - 1, ox + kx - 1): two 16-byte loads, or zeros outside the image; one tap = 4 K-steps.
// Pipeline of a K-step t (12 MFMAs per column block, 8 blocks): blocks 0-2 carry the memory pieces of step t + 1 (9 LDS-DMA
// chunks) and the raw fp32 loads of step t + 2 (16 dwords / 4 x 16 bytes per lane); blocks 3-6 carry the split of the raw
// values of step t + 1 (loaded during step t - 1: ~90 VALU operations, two per MFMA gap -- the slots the matrix instruction
// leaves free) into the fragment registers of the other set; block 7 is held over the barrier.  Round 2's kernel
// (gemm_bf16x3.h A_UNION_FLAT / A_CONV2: one 8-wave workgroup per CU on 32x32x16 blocks, activations through LDS) ran
// these two launches at 0.41 / 0.42 of the engine's roof.
#pragma once
#include "gemm_bf16x3_t16.h"

namespace sttran {

enum { AC_UNION = 1, AC_CONV2 = 2 };

// One K range [ks0, ks0 + nsteps) of the tile (rows m0 .., columns n0 ..) accumulated into `acc`: the activation loads and their
// split, the weight tile's LDS-DMA, the MFMA loop.  Shared by gemm16x3c_kernel (one convolution per launch) and
// pair_conv_fused_x3_kernel (the conv3x3's K range, its ReLU / BN, then the union conv's K range on the same accumulators).
// Returns behind a barrier: every wave is out of the loop, the stage buffers are free.
template <class T, int AKIND>
__device__ __forceinline__ void x3c_kloop(const GemmOperand& A, const FmPlanes& B, int M, int m0, int n0, int ks0, int nsteps,
                                          unsigned char* smem_x3c, f32x4 (&acc)[2][T::NB]) {
  using X = X3T16<T>;
  constexpr int NB = T::NB;
  // (the thread id through an opaque move: the index arithmetic below is recomputed per call and dies with it -- the fused
  //  kernel calls this twice per tile with different operand kinds, see gemm_f32_t16c.h conv_kloop)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  // ---- activation side: this lane's two rows (row block i: row m0 + 32 wave + 16 i + fr), chunk fg = 8 channels ----------
  const float* pu[2];            // A_UNION: &U[pair][8 fg][hw]; A_CONV2: &C2[pair][0][0][8 fg]
  int cy[2], cx[2];              // A_CONV2: input position of tap (0, 0)
  (void)cy; (void)cx;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g0 = m0 + wave * 32 + 16 * i + fr;
    const int g = g0 < M ? g0 : 0;                       // rows past M read pair 0 (never stored)
    const int p = g / kUHW, pos = g - p * kUHW;
    if constexpr (AKIND == AC_UNION) {
      pu[i] = A.ptr + (A.rowoff ? A.rowoff[p] : (int64_t)p * A.ld) + (int64_t)(8 * fg) * kUHW + pos;
    } else {
      const int oy = pos / 7, ox = pos - oy * 7;
      cy[i] = oy - 1; cx[i] = ox - 1;
      pu[i] = A.ptr + (int64_t)p * (128 * kUHW) + 8 * fg;
    }
  }
  float raw[2][2][8];                                     // [buffer][row block][channel of the chunk]
  auto load_raw = [&](int buf, int step) {               // the fp32 values of K-step `step` (relative to ks0)
    const int ks = ks0 + (step < nsteps ? step : 0);      // steps past the range re-read step 0 (never consumed)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (AKIND == AC_UNION) {
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[buf][i][e] = pu[i][((int64_t)ks * kBK + e) * kUHW];
      } else {
        const int tap = ks >> 2, ky = tap / 3, kx = tap - ky * 3;                 // wave-uniform
        const int iy = cy[i] + ky, ix = cx[i] + kx;
        const bool ok = (unsigned)iy < 7u && (unsigned)ix < 7u;
        const float* src = pu[i] + (ok ? (iy * 7 + ix) * 128 : 0) + (ks & 3) * kBK;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { raw[buf][i][e] = ok ? v0[e] : 0.f; raw[buf][i][4 + e] = ok ? v1[e] : 0.f; }
      }
    }
  };
  bf16x8 fa[2][3][2];                                     // [set][plane][row block]
  auto split_half = [&](int buf, int set, int i, int hf) {   // elements 4 hf .. 4 hf + 3 of row block i
    const f32x4 v = {raw[buf][i][4 * hf], raw[buf][i][4 * hf + 1], raw[buf][i][4 * hf + 2], raw[buf][i][4 * hf + 3]};
    bf16x4 h, m, l;
    split3(v, h, m, l);
#pragma unroll
    for (int e = 0; e < 4; ++e) { fa[set][0][i][4 * hf + e] = h[e]; fa[set][1][i][4 * hf + e] = m[e]; fa[set][2][i][4 * hf + e] = l[e]; }
  };

  // ---- weight side: as gemm16x3_kernel ------------------------------------------------------------------------------
  const uint32_t lane_b = (uint32_t)lane * 16u;
  const char* const b_base = reinterpret_cast<const char*>(B.ptr) + ((int64_t)(n0 / 16) * B.kb_total + ks0) * (kFmBlock3 * 2);
  const int64_t b_cb = (int64_t)B.kb_total * (kFmBlock3 * 2);
  auto glds_piece = [&](int q, int step, unsigned char* stage) {
    const int c = min(wave + 4 * q, X::CHUNKS - 1);
    const char* src = b_base + (c / 3) * b_cb + (int64_t)step * (kFmBlock3 * 2) + (c % 3) * 1024;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + lane_b),
                                     (void __attribute__((address_space(3)))*)(stage + c * 1024), 16, 0, 0);
  };

  bf16x8 fb[2][3];
  auto read_b = [&](const unsigned char* stage, int j, int buf) {
#pragma unroll
    for (int p = 0; p < 3; ++p) fb[buf][p] = *reinterpret_cast<const bf16x8*>(stage + (j * 3 + p) * 1024 + lane * 16);
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  auto mma_block = [&](int set, int j, int buf) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][PB[t]], fa[set][PA[t]][i], acc[i][j], 0, 0, 0);
  };

  // prologue: weights of step 0 -> stage 0; raw values of steps 0 and 1; fragments of step 0
#pragma unroll
  for (int q = 0; q < X::CPW; ++q) glds_piece(q, 0, smem_x3c);
  load_raw(0, 0);
  load_raw(1, 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) { split_half(0, 0, i, 0); split_half(0, 0, i, 1); }
  __syncthreads();
  read_b(smem_x3c, 0, 0);

  // K-step t (set = t & 1): MFMAs on fa[set]; raw[set ^ 1] (step t + 1) is split into fa[set ^ 1]; raw[set] (free since the
  // previous step split it) receives step t + 2
  auto k_step = [&](int t, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const unsigned char* cur = smem_x3c + set * X::STAGE_BYTES;
    unsigned char* nxt = smem_x3c + (set ^ 1) * X::STAGE_BYTES;
    const int tn = t + 1 < nsteps ? t + 1 : t;
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) {
      read_b(cur, j + 1, (j + 1) & 1);
      if (j < 3) {                                        // memory pieces: 9 LDS-DMA chunks (3 per block), then the raw loads
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (3 * j + q < X::CPW) glds_piece(3 * j + q, tn, nxt);
        if (j == 2) load_raw(set, t + 2);
      } else {                                            // split of the next step's raw values: one (row block, half) per block
        split_half(set ^ 1, set ^ 1, (j - 3) >> 1, (j - 3) & 1);
      }
      mma_block(set, j, j & 1);
      // 12 MFMAs: [2 MFMA, 1 fragment read, this block's share of the memory pieces / split arithmetic] x 3, then 6 MFMA
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (j < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one LDS-DMA chunk
        else if (j == 2) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // one chunk + a third of the raw loads
        else __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                // a third of the block's split arithmetic
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    read_b(nxt, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_block(set, NB - 1, (NB - 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int t = 0;
    for (; t + 1 < nsteps; t += 2) {
      k_step(t, std::integral_constant<int, 0>{});
      k_step(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < nsteps) k_step(t, std::integral_constant<int, 0>{});
  }
  __syncthreads();
}

// where a lane's two rows (row0, row0 + 16) live in V[p][c][hw]: &V[p][0][hw] (one division by 49 per row and tile)
__device__ __forceinline__ void x3c_vrows(float* V, int C, int M, int row0, float* (&vrow)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = row0 + 16 * i < M ? row0 + 16 * i : 0;
    const int p = r / kUHW;
    vrow[i] = V + (int64_t)p * C * kUHW + (r - p * kUHW);
  }
}

// the finished tile: v = acc + bias (RELU_BN: then ReLU and the eval-mode BN, lib/sttran.py:343-344), stored as 64-byte runs of
// V[p][c][hw] (for one register the 16 lanes of a chunk hold 16 consecutive rows = consecutive hw of one channel)
template <class T, bool RELU_BN>
__device__ __forceinline__ void x3c_store_rows(const float* bias, const float* scale, const float* shift, int M, int row0, int col0,
                                               float* const (&vrow)[2], const f32x4 (&acc)[2][T::NB]) {
  constexpr int NB = T::NB;
  // per column group: the per-channel constants as 16-byte loads, shared by the lane's two rows (the functors' vec() loads
  // them per element: 64 x 1-3 scalar loads per lane and tile, and divides every row by 49)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = col0 + 16 * j;
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RELU_BN) { sc = *reinterpret_cast<const f32x4*>(scale + c); sh = *reinterpret_cast<const f32x4*>(shift + c); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (row0 + 16 * i < M) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e] + b[e];
          if constexpr (RELU_BN) v = relu_nan(v) * sc[e] + sh[e];
          vrow[i][(c + e) * kUHW] = v;
        }
      }
    }
  }
}

template <class T, int AKIND, class Epi>
__global__ void __launch_bounds__(T::NT, 2)
gemm16x3c_kernel(GemmOperand A, FmPlanes B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg, int g_sk,
                 int sk_base, int sk_rem, int half, int tile_base, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3c[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB;
  static_assert(NT == 256 && BM == 128 && NB == 8, "four waves, 32 rows each, 128 columns");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int tiles_dp = dp_per_wg * G;
  const SkRange rg = blk < g_sk ? sk_range(blk, sk_base, sk_rem) : SkRange{0, 0};
  const int pre_end = rg.begin;

  int dp_done = 0;
  for (int it = rg.begin; dp_done < dp_per_wg || it < rg.end;) {
    int tile, ks0, ks1;
    const bool dp = dp_done < dp_per_wg && !(it < pre_end);
    if (dp) {
      tile = dp_done * G + blk;
      ks0 = 0; ks1 = ksteps;
      ++dp_done;
    } else {
      const int t = it / ksteps;
      tile = tiles_dp + t;
      ks0 = it - t * ksteps;
      ks1 = min(ksteps, ks0 + (rg.end - it));
    }
    tile += tile_base;                                       // a launch over the tiles behind pair_conv_fused_x3_kernel's
    const int nsteps = ks1 - ks0;
    int tile_m, tile_n;
    tile_origin_rt(tile, tiles_m, tiles / tiles_m, half, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int row0 = m0 + wave * 32 + fr;
    const int col0 = n0 + 4 * fg;
    float* vrow[2];
    x3c_vrows(epi.V, epi.C, M, row0, vrow);
    if constexpr (EpiInit<Epi>::value) {
      // C += A B (EpiUnionRows): the K range that starts a tile accumulates onto the output's old values
      if (ks0 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (row0 + 16 * i < M) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j][e] = vrow[i][(col0 + 16 * j + e) * kUHW];
          }
        }
      }
    }
    x3c_kloop<T, AKIND>(A, B, M, m0, n0, ks0, nsteps, smem_x3c, acc);
    if (nsteps == ksteps) {
      if constexpr (AKIND == AC_CONV2) x3c_store_rows<T, true>(epi.bias, epi.scale, epi.shift, M, row0, col0, vrow, acc);
      else x3c_store_rows<T, false>(epi.bias, nullptr, nullptr, M, row0, col0, vrow, acc);
    } else {
      f32x4* sp = reinterpret_cast<f32x4*>(slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN)) + tid;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) sp[(i * NB + j) * NT] = acc[i][j];
    }
    if (!dp) it += nsteps;
  }
}


// ---- both convolutions of a pair block in ONE pass over the tile (round 6; the exact engine's form: gemm_f32_t16c.h
//      pair_conv_fused_kernel) -- the whole rounds of a launch only (the ReLU between the two K ranges makes a tile indivisible
//      for stream-K); the leftover tiles go through the two launches above with `tile_base`
template <class T>
__global__ void __launch_bounds__(T::NT, 2)
pair_conv_fused_x3_kernel(GemmOperand A2, FmPlanes B2, GemmOperand A1, FmPlanes B1, int M, int K1, int tiles_m, int tiles, int ntiles,
                          int half, EpiConvRows e2, EpiUnionRows e1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3c[];
  constexpr int BM = T::BM, BN = T::BN, NB = T::NB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  for (int tile = blk; tile < ntiles; tile += G) {         // whole tiles only; the last round may be partly filled
    int tile_m, tile_n;
    tile_origin_rt(tile, tiles_m, tiles / tiles_m, half, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    x3c_kloop<T, AC_CONV2>(A2, B2, M, m0, n0, 0, 1152 / kBK, smem_x3c, acc);
    const int col0 = n0 + 4 * fg;
#pragma unroll
    for (int j = 0; j < NB; ++j) {                           // conv3x3's bias, ReLU, eval-mode BN on the accumulators
      const int c = col0 + 16 * j;
      const f32x4 b = *reinterpret_cast<const f32x4*>(e2.bias + c), sc = *reinterpret_cast<const f32x4*>(e2.scale + c),
                  sh = *reinterpret_cast<const f32x4*>(e2.shift + c);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = relu_nan(acc[i][j][e] + b[e]) * sc[e] + sh[e];
    }
    x3c_kloop<T, AC_UNION>(A1, B1, M, m0, n0, 0, K1 / kBK, smem_x3c, acc);
    const int row0 = m0 + wave * 32 + fr;
    float* vrow[2];
    x3c_vrows(e1.V, e1.C, M, row0, vrow);
    x3c_store_rows<T, false>(e1.bias, nullptr, nullptr, M, row0, col0, vrow, acc);
  }
}

}  // namespace sttran
