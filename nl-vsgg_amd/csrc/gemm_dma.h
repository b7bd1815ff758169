// gemm_dma.h -- second-generation exact-fp32 MFMA GEMM of the nn.Linear layers (gfx950 / CDNA4).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      A rows optionally gathered; operands meet the B_KMAJOR_PAD contract
//
// Same arithmetic and the same k order inside every accumulator as gemm_f32_mfma.h (v_mfma_f32_32x32x2_f32, one
// ds_read_b128 feeding four MFMAs); what changed is everything around the MFMAs:
//   * staging by LDS-DMA (`global_load_lds_dwordx4`): global -> LDS without passing through VGPRs, no ds_write, no
//     staging registers.  One wave-instruction moves 8 rows x 128 bytes.  LDS rows are UNPADDED (128 bytes) because a
//     DMA destination is wave-uniform base + lane * 16; the ds_read_b128 fragment reads stay conflict-free through an
//     XOR swizzle of the 16-byte chunk index by (row >> 1) & 7, applied on the SOURCE side (each lane fetches the
//     chunk that belongs at its LDS position) and again when the fragments are read;
//   * the MFMA operands are swapped (weights feed the "A" port, activations the "B" port), so a lane's 16
//     accumulator registers hold, for ONE output row, four groups of four CONSECUTIVE columns: the epilogue loads
//     bias / residual and stores C as 16-byte vectors (4x fewer memory instructions than the column-per-lane form);
//   * 128x128 tiles run two workgroups per CU (64 KB of LDS each) and the second-dispatched half of the grid walks
//     its work in the opposite order (stream-K range first, whole tiles after), so the two workgroups of a CU reach
//     their epilogues at different times: one keeps the matrix pipe busy while the other stores.
// The schedule is the hybrid data-parallel + stream-K one of gemm_f32_mfma.h (same SkRange helpers, same
// deterministic fix-up in ascending workgroup order).
#pragma once
#include "gemm_f32_mfma.h"

namespace sttran {

#define STTRAN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define STTRAN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BM_, int BN_, int WM_, int WN_>
struct DmaTile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int ROWS = BM + BN;            // staged rows per K-step: A rows, then B rows
  static constexpr int IPW = ROWS / 8 / NW;       // DMA wave-instructions per wave per K-step (8 rows each)
  static constexpr int STAGE = ROWS * kBK;        // floats per LDS stage (unpadded 128-byte rows)
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int GROUP_N = 8;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
  static_assert(ROWS % (8 * NW) == 0, "DMA row groups must divide evenly over the waves");
  static_assert(BM % 16 == 0, "the swizzle of a B row must not depend on BM");
};

// ORDER: 0 = whole tiles first, then the stream-K range; 1 = the other way round for the second-dispatched half of
// the grid (blockIdx.x >= half), see the file comment.
template <class T, class Epi>
__global__ void __launch_bounds__(T::NT)
gemm_dma_kernel(GemmOperand A, GemmOperand B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg,
                int g_sk, int sk_base, int sk_rem, int half, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN, IPW = T::IPW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int fr = lane & 31, fh = lane >> 5;
  // fragment read offsets (floats) inside a stage: row * 32 + 4 * ((2 kb + fh) ^ swz), swz = (row >> 1) & 7 = (fr >> 1) & 7
  const int swz = (fr >> 1) & 7;
  int a_off[4], b_off[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int ko = 4 * ((2 * kb + fh) ^ swz);
    a_off[kb] = (wm * (BM / T::WM) + fr) * kBK + ko;
    b_off[kb] = (BM + wn * (BN / T::WN) + fr) * kBK + ko;
  }
  // DMA lanes: instruction i of this wave fills stage rows 8 g .. 8 g + 7, g = wave * IPW + i; lane l writes LDS chunk
  // (l & 7) of row 8 g + (l >> 3), which must hold global chunk (l & 7) ^ swz(row)
  const int drow = lane >> 3;

  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int tiles_dp = dp_per_wg * G;
  const SkRange rg = blk < g_sk ? sk_range(blk, sk_base, sk_rem) : SkRange{0, 0};
  const bool sk_first = (int)blockIdx.x >= half;

  int dp_done = 0;
  for (int it = rg.begin; dp_done < dp_per_wg || it < rg.end;) {
    int tile, ks0, ks1;
    const bool dp = sk_first ? !(it < rg.end) : dp_done < dp_per_wg;
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
    tile_origin<T::GROUP_N>(tile, tiles_m, tiles / tiles_m, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // per-lane source pointers of this tile's DMA pieces (rows past M / N are clamped to row 0: finite data that
    // only reaches accumulators the epilogue never stores)
    const float* src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int row = (wave * IPW + i) * 8 + drow;                  // row inside the stage
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      if (row < BM) {
        const int g = m0 + row;
        src[i] = A.ptr + (int64_t)(g < M ? (A.rowidx ? A.rowidx[g] : g) : 0) * A.ld + ks0 * kBK + chunk * 4;
      } else {
        const int g = n0 + row - BM;
        src[i] = B.ptr + (int64_t)(g < N ? g : 0) * B.ld + ks0 * kBK + chunk * 4;
      }
    }
    auto dma = [&](int i, float* stage, int step) {
      __builtin_amdgcn_global_load_lds(STTRAN_GPTR(src[i] + step * kBK), STTRAN_LPTR(stage + (wave * IPW + i) * 8 * kBK), 16, 0, 0);
    };
    auto read_frags = [&](const float* stage, int kb, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(stage + a_off[kb] + i * 32 * kBK);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(stage + b_off[kb] + j * 32 * kBK);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // swapped operand ports: D[n][m] -- the lane's column index (lane & 31) runs over M, the register index over N
    auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][e], fa[i][e], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: K-step 0 into stage 0
#pragma unroll
    for (int i = 0; i < IPW; ++i) dma(i, smem, 0);
    __syncthreads();                       // (vmcnt(0) + barrier: hipcc drains LDS-DMA in front of a barrier)
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    read_frags(smem, 0, fa0, fb0);
    constexpr int NM = 4 * TM * TN;
    for (int t = 0; t < nsteps; ++t) {
      const float* cur = smem + (t & 1) * T::STAGE;
      float* nxt = smem + ((t + 1) & 1) * T::STAGE;
      // the step past the range re-reads step 0 into the idle stage (no branch in the loop)
      const int tn = (t + 1 < nsteps) ? t + 1 : 0;
      read_frags(cur, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      {
        // group 0: the DMA of the next K-step rides in the MFMA gaps
        int n = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0[j][e], fa0[i][e], acc[i][j], 0, 0, 0);
              if (n < IPW) dma(n, nxt, tn);
              ++n;
            }
#pragma unroll
        for (; n < IPW; ++n) dma(n, nxt, tn);
#pragma unroll
        for (int q = 0; q < IPW && q < NM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 2, fa0, fb0);
      mma(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 3, fa1, fb1);
      mma(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                     // next stage landed (vmcnt(0)) and every wave is done reading `cur`
      read_frags(nxt, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);   // issue these reads BEFORE the held-over MFMA group, which then hides them
      mma(fa1, fb1);
    }

    // C/D layout with swapped ports: row m = lane & 31 of block i, cols n = 8 q + 4 (lane >> 5) + {0..3} of block j
    const int row = m0 + wm * (BM / T::WM) + fr;
    const int cbase = n0 + wn * (BN / T::WN) + 4 * fh;
    if (nsteps == ksteps) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = row + i * 32;
        if (r < M) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = cbase + j * 32 + 8 * q;
              const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
              if (col + 3 < N) epi.vec(r, col, v);
              else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                  if (col + c < N) epi(r, col + c, v[c]);
              }
            }
        }
      }
    } else {
      // partial K range: park the raw accumulators as 16-byte vectors, register-major (1 KB per wave-instruction)
      f32x4* sp = reinterpret_cast<f32x4*>(slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN)) + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            sp[((i * TN + j) * 4 + q) * NT] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    }
    if (!dp) it += nsteps;
    __syncthreads();                       // the next tile's prologue DMA overwrites stage 0
  }
}

}  // namespace sttran
