// gemm_bf16x3_t16.h -- fp32 GEMM emulated on the bf16 matrix pipe, second generation (round 6): the structure of
// gemm_f32_t16.h (4-wave workgroups, two per CU, a wave owns 32 rows x all columns of a 128 x 176 / 128 x 128 tile) on
// v_mfma_f32_16x16x32_bf16, with BOTH operands arriving pre-split.  SURVEY.md 7: "fp32 MFMA ... or 3 x bf16 split" -- opt-in
// (`sttran_set_gemm_engine`), never the default, never the bench line's `value`.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),   x = x1 + x2 + x3 (bf16 planes, gemm_bf16x3.h), six cross products per term
//
// What is different from gemm_bf16x3.h (round 2: 256 x 128 tile, one 8-wave workgroup per CU, 0.42 of the engine's roof):
//   * ACTIVATIONS ARE SPLIT ONCE, where they are produced (`split_fm_kernel` here; fused into the producing kernels where
//     that kernel has the values in registers), not by every tile column's A loader -- round 2's loader split an A element
//     N / 128 times (16 x for N = 1936) at ~5 VALU operations each, 14 % of the kernel;
//   * both operands live in HBM FRAGMENT-MAJOR: a [16 rows] x [32 k] block of one plane is 1 KB in exactly the order the
//     MFMA wants it -- lane l = (row l % 16, k chunk l / 16) holds 8 consecutive k --, the three planes of a block back to
//     back (3 KB).  Element (r, k) of plane p:  ((r / 16 * KB + k / 32) * 3 + p) * 512 + ((r % 16) + 16 * (k % 32 / 8)) * 8 + k % 8.
//     So a fragment is ONE coalesced 1 KB load, an LDS image is a straight copy (no swizzle arithmetic: `ds_read_b128` of a
//     lane-linear 1 KB block is conflict-free by construction of the instruction's lane groups), and a weight tile's K-step
//     goes global -> LDS without passing registers (`global_load_lds_dwordx4`, one instruction per 1 KB block).
//     That lane-major order is the WEIGHTS' (they pass through LDS).  The ACTIVATION planes use the same blocks with the 1 KB
//     of a plane ROW-major ([16 rows][32 k], 64 bytes per row: element (r, k) at (r % 16) * 32 + k % 32): their fragments go
//     straight to registers, where any order inside the 1 KB is the same coalesced load, and a producer that holds a ROW
//     (LayerNorm, a GEMM epilogue) writes 32-64 contiguous bytes per plane instead of 8 (measured: the lane-major
//     LayerNorm variant took 130 us against 47 us without planes -- every 8-byte store its own 32-byte sector);
//   * the A fragments never touch LDS: wave w owns rows 32 w .. 32 w + 31 of the tile and ALL its columns, so no other wave
//     needs them -- six 1 KB loads per wave and K-step, straight into the registers the MFMAs read, one K-step ahead.
//     LDS holds the weight tile only: 2 stages x (BN / 16) x 3 KB = 66 KB (176 columns), two workgroups per CU;
//   * 176 = 11 blocks of 16 columns divides N = 1936 / 3872 / 5808 exactly (round 2's 128-column tile padded 1936 to 2048).
// Per K-step and wave: 6 x 2 x NB MFMAs (132 at NB = 11, 16 cycles each), 3 NB `ds_read_b128`, 6 global loads, 8-9 LDS-DMA
// loads; one barrier.  Accumulator layout, epilogue, stream-K schedule, parking format and fix-up launch are gemm16_kernel's
// (same Tile16 traits: `gemm16_fixup_kernel` serves both).
#pragma once
#include <type_traits>

#include "gemm_bf16x3.h"
#include "gemm_f32_t16.h"

namespace sttran {

constexpr int kFmBlock = 512;                 // bf16 elements of one plane of one [16][32] block
constexpr int kFmBlock3 = 3 * kFmBlock;       // the three planes of a block, back to back

// Fragment-major planes of an operand: base of block (0, 0); kb_total = K blocks per row block (ceil(K / 32))
struct FmPlanes {
  const __bf16* ptr;
  int kb_total;
  int rb_total;                               // row blocks that exist (reads are clamped to them)
};

// fp32 rows -> fragment-major planes.  One wave per [16][32] block: lane (r, c) reads 8 floats of row 16 rb + r at k = 32 kb +
// 8 c (two 16-byte loads: the 16 rows' 128-byte lines), splits them and writes 16 bytes per plane at lane * 16 of the
// block's three 1 KB images (fully coalesced).  Rows >= M and columns >= K are written as ZERO (the K tail of the last
// block must not carry anything into the dot product; tail rows are computed by the GEMM and never stored).
// Row r of the operand: src + rowoff[r] (64-bit element offsets), else src + rowidx[r] * ld, else src + r * ld.
static __global__ void __launch_bounds__(256)
split_fm_kernel(const float* __restrict__ src, int64_t ld, const int32_t* __restrict__ rowidx, const int64_t* __restrict__ rowoff,
                int M, int K, int kb_total, int64_t blocks, __bf16* __restrict__ planes, int row_major) {
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= blocks) return;
  const int rb = (int)(blk / kb_total), kb = (int)(blk - (int64_t)rb * kb_total);
  const int r = rb * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
  if (r < M) {
    const float* row = src + (rowoff ? rowoff[r] : (int64_t)(rowidx ? rowidx[r] : r) * ld) + k0;
    if (k0 + 8 <= K) { v0 = *reinterpret_cast<const f32x4*>(row); v1 = *reinterpret_cast<const f32x4*>(row + 4); }
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { if (k0 + e < K) v0[e] = row[e]; if (k0 + 4 + e < K) v1[e] = row[4 + e]; }
    }
  }
  bf16x4 h0, m0, l0, h1, m1, l1;
  split3(v0, h0, m0, l0);
  split3(v1, h1, m1, l1);
  // weights: lane-major (lane * 16 bytes); activations: row-major inside the block (row * 64 + chunk * 16 bytes)
  __bf16* out = planes + blk * kFmBlock3 + (row_major ? (lane & 15) * 32 + (lane >> 4) * 8 : lane * 8);
  *reinterpret_cast<bf16x8*>(out) = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  *reinterpret_cast<bf16x8*>(out + kFmBlock) = bf16x8{m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
  *reinterpret_cast<bf16x8*>(out + 2 * kFmBlock) = bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

// bf16 element offset of (row, col) in plane 0 of ACTIVATION planes (row-major blocks) with kb_total K blocks per row block
__device__ __forceinline__ int64_t fm_offset(int row, int col, int kb_total) {
  return ((int64_t)(row >> 4) * kb_total + (col >> 5)) * kFmBlock3 + ((row & 15) << 5) + (col & 31);
}
// four consecutive columns (col % 4 == 0) of one row -> the three planes, 8 bytes each
__device__ __forceinline__ void fm_store4(__bf16* planes, int kb_total, int row, int col, const f32x4& v) {
  bf16x4 h, m, l;
  split3(v, h, m, l);
  __bf16* o = planes + fm_offset(row, col, kb_total);
  *reinterpret_cast<bf16x4*>(o) = h;
  *reinterpret_cast<bf16x4*>(o + kFmBlock) = m;
  *reinterpret_cast<bf16x4*>(o + 2 * kFmBlock) = l;
}

// Epilogue of a GEMM whose output is ONLY the next GEMM's activation operand (linear1 -> ReLU -> linear2,
// lib/transformer.py:24,27,53,56): out = act(acc + bias) leaves as fragment-major planes -- 6 bytes per element instead of
// 4 for an fp32 row nobody else reads, and no split pass in front of the next launch.
struct EpiActPlanes {
  static constexpr bool kVector = true;
  const float* bias; __bf16* planes; int kb_total; int relu;
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
    v += *reinterpret_cast<const f32x4*>(bias + col);
    if (relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = relu_nan(v[c]);
    }
    fm_store4(planes, kb_total, row, col, v);
  }
};

// Epilogue of linear2 of a decoder layer whose output the NEXT layer projects (lib/transformer.py:56-58 -> :51): C = acc + bias +
// residual goes out as the fp32 row the residual path and the gathers need AND as the next in_proj's activation planes.
struct EpiResPlanes {
  static constexpr bool kVector = true;
  float* C; int64_t ldc; const float* bias; const float* res; int64_t ldres; __bf16* planes; int kb_total;
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
    v += *reinterpret_cast<const f32x4*>(bias + col);
    v += *reinterpret_cast<const f32x4*>(res + (int64_t)row * ldres + col);
    *reinterpret_cast<f32x4*>(C + (int64_t)row * ldc + col) = v;
    fm_store4(planes, kb_total, row, col, v);
  }
};

// T = Tile16<128, 176> or Tile16<128, 128> (gemm_f32_t16.h: only BM, BN, NB, NT are used)
template <class T>
struct X3T16 {
  static constexpr int NB = T::NB;
  static constexpr int STAGE_BYTES = NB * 3 * 1024;        // the weight tile's K-step: NB column blocks x 3 planes x 1 KB
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int CHUNKS = NB * 3;                    // 1 KB LDS-DMA pieces per K-step
  static constexpr int CPW = (CHUNKS + 3) / 4;             // ... per wave (the last wave(s) have one less)
};

// ABL: timing-only ablations for tools/x3_bench.py (experiment builds, wrong results): 1 = no epilogue stores, 2 = no barrier in
// the K loop, 3 = no LDS-DMA in the loop (the weight stage is never refilled), 4 = no A-fragment loads in the loop, 5 = 3 + 4,
// 6 = every load of the loop issued but from ONE small footprint (row blocks 0..7, column blocks 0..10, K block 0: always an L2
// hit) -- separates the cost of the load path from the cost of the misses.  (Round 6, one box: [21120,1936,1936] 838 us; no
// epilogue stores 811; no barrier 830; no LDS-DMA 762; no A loads 687; neither 592; all loads but always hits 741.  Non-temporal
// hints on the A loads / the LDS-DMA / both: 862 / 931 / 993 us -- the caches DO serve neighbours; removed again.)
template <class T, class Epi, int ABL = 0>
__global__ void __launch_bounds__(T::NT, 2)
gemm16x3_kernel(FmPlanes A, FmPlanes B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg, int g_sk,
                int sk_base, int sk_rem, int half, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
  using X = X3T16<T>;
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB;
  static_assert(NT == 256 && BM == 128, "four waves, 32 rows each");
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
    const int nsteps = ks1 - ks0;
    int tile_m, tile_n;
    tile_origin_rt(tile, tiles_m, tiles / tiles_m, half, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // Operand addresses = a wave-UNIFORM 64-bit base (scalar registers: `wave` is a readfirstlane value) + a 32-bit per-lane
    // byte offset: one VGPR for all of A's fragments and one for all LDS-DMA pieces, instead of a 64-bit pointer per piece
    // (9 + 2 pointers = 22 VGPRs: with them the 176-column tile spilled).
    // A: this wave's two row blocks (clamped: a tail block re-reads the last one, its rows are never stored), row-major
    // blocks; B: the weight tile's column blocks; per K-step both advance by one [16][32] block = 3 KB.
    const uint32_t lane_a = (uint32_t)(((lane & 15) * 32 + (lane >> 4) * 8) * 2), lane_b = (uint32_t)lane * 16u;
    const char* a_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rb = ABL == 6 ? 2 * wave + i : min(m0 / 16 + 2 * wave + i, A.rb_total - 1);
      a_base[i] = reinterpret_cast<const char*>(A.ptr) + ((int64_t)rb * A.kb_total + (ABL == 6 ? 0 : ks0)) * (kFmBlock3 * 2);
    }
    const char* const b_base = reinterpret_cast<const char*>(B.ptr) + ((int64_t)(ABL == 6 ? 0 : n0 / 16) * B.kb_total + (ABL == 6 ? 0 : ks0)) * (kFmBlock3 * 2);
    const int64_t b_cb = (int64_t)B.kb_total * (kFmBlock3 * 2);        // bytes between column blocks
    // LDS-DMA piece q of this wave: chunk c = wave + 4 q of the stage image [NB][3][1 KB] = (column block c / 3, plane c % 3)
    // (33 chunks over 4 waves x 9 pieces: the three pieces past the end re-stage the last chunk -- same bytes to the same
    //  address -- instead of sitting under a wave-uniform branch: a branch ends the scheduling region, and with one around
    //  every piece the fragment reads of block j + 1 were issued, and waited for, in FRONT of block j's MFMAs)
    auto glds_piece = [&](int q, int step, unsigned char* stage) {
      const int c = min(wave + 4 * q, X::CHUNKS - 1);
      const char* src = b_base + (c / 3) * b_cb + (int64_t)(ABL == 6 ? 0 : step) * (kFmBlock3 * 2) + (c % 3) * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + lane_b),
                                       (void __attribute__((address_space(3)))*)(stage + c * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int step, unsigned char* stage) {          // K-step `step` of the weight tile -> `stage`
#pragma unroll
      for (int q = 0; q < X::CPW; ++q) glds_piece(q, step, stage);
    };
    bf16x8 fa[2][3][2];                                             // [set][plane][row block]
    auto load_a1 = [&](int set, int step, int i, int pl) {
      fa[set][pl][i] = *reinterpret_cast<const bf16x8*>(a_base[i] + (int64_t)(ABL == 6 ? 0 : step) * (kFmBlock3 * 2) + pl * 1024 + lane_a);
    };
    auto load_a = [&](int set, int step) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) load_a1(set, step, i, pl);
    };

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fb[2][3];
    auto read_b = [&](const unsigned char* stage, int j, int buf) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[buf][p] = *reinterpret_cast<const bf16x8*>(stage + (j * 3 + p) * 1024 + lane * 16);
    };
    // the six cross terms of one (row block pair, column block), small ones first; two accumulators alternate
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    auto mma_block = [&](int set, int j, int buf) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)      // weights on the "A" port: a lane holds 4 consecutive columns of one output row
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][PB[t]], fa[set][PA[t]][i], acc[i][j], 0, 0, 0);
    };
    // one memory piece of the NEXT K-step: pieces 0 .. CPW-1 = this wave's LDS-DMA chunks, CPW .. CPW+5 = its A fragments
    constexpr int NVM = X::CPW + 6;
#ifndef X3_PPB
#define X3_PPB 4
#endif
    // memory pieces per block: 4 puts the 15 (128 columns: 12) pieces of a step into its first 4 (3) blocks, so the last one is
    // issued 6 (4) blocks in front of the barrier that waits for it (2 per block, the first form: the last piece 2 blocks / 1
    // block in front of it) -- same-box A/B (tools/experiments/x3_variants.sh): 2 / 3 / 4 / 5 per block = 841 / 835 / 832 / 838 us
    // on [21120,1936,1936], 2 121 / 2 079 / 2 081 / 2 096 on [21120,5808,1936], 887 / 875 / 869 / 880 on [21120,2048,1936]
    constexpr int PPB = X3_PPB;
    auto vmem_piece = [&](int n, int set_next, int step, unsigned char* stage) {
      if (n < X::CPW) { if constexpr (ABL != 3 && ABL != 5) glds_piece(n, step, stage); }
      else if (n < NVM) { if constexpr (ABL != 4 && ABL != 5) load_a1(set_next, step, (n - X::CPW) / 3, (n - X::CPW) % 3); }
    };

    stage_b(0, smem_x3);
    load_a(0, 0);
    __syncthreads();                              // (hipcc drains the LDS-DMA with vmcnt(0) in front of the barrier)
    read_b(smem_x3, 0, 0);

    // One K-step.  On entry fb[par] holds column block 0's fragments (read behind the previous barrier).  Block j runs its 12
    // MFMAs with the 3 fragment reads of block j + 1 and PPB memory pieces of the NEXT K-step between them (a piece = one 1 KB
    // LDS-DMA chunk of the weight tile or one A fragment: 14-15 per wave, spread over the first blocks -- issued in one burst
    // at the top of the step they held the wave for ~300 cycles before its first MFMA).  The LAST block is held over the
    // barrier: its MFMAs run after the next step's first fragment reads have been issued and cover the barrier wait and
    // their latency (gemm16_kernel does the same).  PAR = the fragment buffer of block 0: with an odd number of column
    // blocks (11) consecutive steps start on alternating buffers.
    auto k_step = [&](int t, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      constexpr int par = (NB & 1) ? set : 0;
      const unsigned char* cur = smem_x3 + set * X::STAGE_BYTES;
      unsigned char* nxt = smem_x3 + (set ^ 1) * X::STAGE_BYTES;
      const int tn = t + 1 < nsteps ? t + 1 : t;                   // the last step re-loads itself (never consumed)
#pragma unroll
      for (int j = 0; j + 1 < NB; ++j) {
        read_b(cur, j + 1, (par + j + 1) & 1);
#pragma unroll
        for (int p = 0; p < PPB; ++p) vmem_piece(PPB * j + p, set ^ 1, tn, nxt);
        mma_block(set, j, (par + j) & 1);
        constexpr int GR = PPB > 3 ? PPB : 3;                       // groups of 2 MFMAs that carry a fragment read / a memory piece
#pragma unroll
        for (int r = 0; r < GR; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (r < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (r < PPB && PPB * j + r < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if constexpr (12 - 2 * GR > 0) __builtin_amdgcn_sched_group_barrier(0x008, 12 - 2 * GR, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_assert(PPB * (NB - 1) >= NVM && PPB <= 6, "the next K-step's memory pieces must fit the blocks in front of the held-over one");
      if constexpr (ABL != 2) __syncthreads();    // every wave has read `cur`; `nxt` and the next A fragments have landed
      read_b(nxt, 0, (par + NB) & 1);
      __builtin_amdgcn_sched_barrier(0);          // issue these reads BEFORE the held-over block, which then hides them
      mma_block(set, NB - 1, (par + NB - 1) & 1);
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
    __syncthreads();                              // the next item's prologue overwrites stage 0: every wave is out of the loop

    // ---- epilogue: gemm16_kernel's (same accumulator layout: row = lane % 16 of block i, columns 4 (lane / 16) + {0..3} of block j)
    const int row0 = m0 + wave * 32 + fr;
    const int col0 = n0 + 4 * fg;
    if constexpr (ABL == 1) {
      if (acc[0][0][0] == 12345.678f) slab[tid] = acc[1][NB - 1][3];      // keep the accumulators alive; never true
    } else
    if constexpr (std::is_same<Epi, EpiActPlanes>::value) {
      if (nsteps == ksteps) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = row0 + 16 * i;
          if (r < M) {
#pragma unroll
            for (int j = 0; j < NB; ++j) epi.vec(r, col0 + 16 * j, acc[i][j]);
          }
        }
      }
    } else if constexpr (std::is_same<Epi, EpiResPlanes>::value) {
      if (nsteps == ksteps) {
        // all loads of a row first (bias + residual), then its stores: a load behind a store waits for the store's
        // acknowledgement (gemm16_kernel's strip epilogue, same reason)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = row0 + 16 * i;
          if (r < M) {
            f32x4 w[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
              w[j] = acc[i][j] + *reinterpret_cast<const f32x4*>(epi.bias + col0 + 16 * j) +
                     *reinterpret_cast<const f32x4*>(epi.res + (int64_t)r * epi.ldres + col0 + 16 * j);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              *reinterpret_cast<f32x4*>(epi.C + (int64_t)r * epi.ldc + col0 + 16 * j) = w[j];
              fm_store4(epi.planes, epi.kb_total, r, col0 + 16 * j, w[j]);
            }
          }
        }
      }
    } else
    if (nsteps == ksteps) {
      const int rows2[2] = {row0, row0 + 16};
      const bool valid2[2] = {row0 < M, row0 + 16 < M};
      constexpr int H0R = NB <= 8 ? NB : (NB + 1) / 2;
      bool done = epi_linear_rows2<NB, 0, H0R>(epi.e, rows2, valid2, col0, acc);
      if constexpr (H0R < NB) {
        if (done) epi_linear_rows2<NB, H0R, NB - H0R>(epi.e, rows2, valid2, col0, acc);
      }
      if (done) {
      } else if constexpr (NB <= 8) {
        int cols[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) cols[j] = col0 + 16 * j;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = row0 + 16 * i;
          if (r < M) epi_linear_strip<NB>(epi.e, r, cols, acc[i]);
        }
      } else {
        constexpr int H0 = (NB + 1) / 2, H1 = NB - H0;
        int c0[H0], c1[H1];
#pragma unroll
        for (int j = 0; j < H0; ++j) c0[j] = col0 + 16 * j;
#pragma unroll
        for (int j = 0; j < H1; ++j) c1[j] = col0 + 16 * (H0 + j);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = row0 + 16 * i;
          if (r < M) {
            f32x4 a0[H0], a1[H1];
#pragma unroll
            for (int j = 0; j < H0; ++j) a0[j] = acc[i][j];
#pragma unroll
            for (int j = 0; j < H1; ++j) a1[j] = acc[i][H0 + j];
            epi_linear_strip<H0>(epi.e, r, c0, a0);
            epi_linear_strip<H1>(epi.e, r, c1, a1);
          }
        }
      }
    }
    if (nsteps != ksteps) {
      // partial K range: park the raw accumulators in gemm16_kernel's format (gemm16_fixup_kernel sums them)
      f32x4* sp = reinterpret_cast<f32x4*>(slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN)) + tid;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) sp[(i * NB + j) * NT] = acc[i][j];
    }
    if (!dp) it += nsteps;
  }
}

}  // namespace sttran
