// gemm_f32_t16.h -- exact-fp32 MFMA GEMM on 16x16 blocks: the 128 x 176 tile of the N = 1936 family.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  N % 176 == 0,  both operands on the padded K-major contract (B_KMAJOR_PAD)
//
// Why a second tile family (lib/transformer.py:9-12,38-42: nn.MultiheadAttention in/out projections and linear2 all have
// N = 1936, 3872 or 5808 = 11, 22, 33 x 176):
//   * 1936 = 16 x 121 has no divisor that is a multiple of 32, so the 32x32x2 tiles of gemm_f32_mfma.h pad it to 2048
//     (5.8 % of those GEMMs' MFMAs compute columns nobody stores); 176 = 11 blocks of v_mfma_f32_16x16x4_f32 divides it
//     exactly -- same 64 FLOP/clk/SIMD as the 32x32x2 form;
//   * the 256 x 128 tile owns a CU alone (111 KB of LDS, 8 waves = 2 per SIMD in ONE workgroup): every wave of the CU is
//     in the prologue / epilogue of a tile at the same time and the matrix pipe idles meanwhile (~7 % of the kernel, PMC).
//     Here a workgroup is 4 waves (one per SIMD) on 128 x 176 with 76 KB of LDS, so TWO independent workgroups share a
//     CU: while one stores its tile and fills its pipeline, the other one's MFMAs keep the pipe busy.
//
// Wave w owns rows 32 w .. 32 w + 31 and ALL 176 columns: 2 x 11 accumulators of 4 registers (88 VGPRs).  The weights
// feed the MFMA's "A" port and the activations its "B" port (as in the swapped-port form of gemm_f32_mfma.h), so a lane
// holds, per accumulator, four CONSECUTIVE columns of one output row: bias / residual / C are 16-byte accesses.
// LDS: rows of 32 floats (BK = 32), unpadded, the 16-byte slot index XOR-ed with (row >> 1) & 7 -- ds_read_b128 of a
// 16-row fragment (lane = (row l % 16, k-chunk l / 16)) is then conflict-free in each of the instruction's four
// 16-lane groups, and so are the 8-lanes-per-row ds_write_b128 of the staging.  One ds_read_b128 feeds four MFMAs
// (element e of both fragments = k 16 kb + 4 (l / 16) + e: the same permutation of k for both operands).
// Global -> VGPR -> LDS staging one K-step ahead: the loads of step t + 1 are issued behind the first MFMAs of step t and
// written to the other stage behind its last ones (~0.8 K-step = 4 us in flight; the co-resident workgroup covers the rest).
// Schedule: the hybrid data-parallel + stream-K plan of gemm_f32_mfma.h (same SkRange / parking / fix-up scheme).
#pragma once
#include <type_traits>

#include "gemm_f32_mfma.h"

namespace sttran {

#ifdef STTRAN_GEMM_EXPERIMENT
// phase clocks of the 128 x 176 kernel (experiment builds, STTRAN_T16_ABLATE=9): summed over workgroups, s_memtime ticks
// [0] prologue (tile start -> first MFMA), [1] main loop, [2] epilogue / parking, [3] tiles, [4] K-steps
__device__ unsigned long long g_t16_clk[8];
// STTRAN_T16_SKEW (experiment): start-up skew of the workgroups that share operand panels inside an XCD, in units of
// s_sleep(16) ~ 1 024 clocks: workgroup blk sleeps ((blk & 7) + ((blk >> 3) & 7)) * skew units before its first tile, so that
// the 8 sharers of an A panel (consecutive blk) and the 8 sharers of a W panel (blk 8 apart) ask for a line ~0.4 us apart
// instead of within the same microsecond.  Tests the hit-on-miss explanation of the GEMM class's fabric reads.
__device__ int g_t16_skew;
// ABL == 9 trace ("what does the resident partner do during the other's epilogue?"): one record of 8
// 64-bit words per tile segment -- HW_ID, XCC_ID, blk, tile, and the four s_memtime stamps (tile start, first MFMA, last
// MFMA, epilogue stores acknowledged) -- into a caller buffer (sttran_debug_t16_trace); 100 MHz chip-wide clock
__device__ unsigned long long* g_t16_trace;
__device__ unsigned int g_t16_trace_cap, g_t16_trace_n;
#endif

template <int BM_, int BN_>
struct Tile16 {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int WAVES = BM / 32, NT = WAVES * 64;
  static constexpr int NB = BN / 16;                       // 16-column blocks per wave (all of the tile's columns)
  static constexpr int ROWS = BM + BN;                     // staged rows per K-step
  static constexpr int STAGE = ROWS * kBK;                 // floats per LDS stage (unpadded 128-byte rows)
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int AV = BM * 8 / NT;                   // 16-byte pieces per thread per K-step, A rows
  static constexpr int BVF = (BN * 8) / NT;                // ... B rows: full rounds
  static constexpr int BREM = BN * 8 - BVF * NT;           // ... and the threads of the last, partial round
  static constexpr int BV = BVF + (BREM ? 1 : 0);
  static constexpr int GROUP_N = 8;
  static constexpr int RPR = NT / 8;                       // rows staged per round (8 threads per 128-byte row)
  static_assert(BM % 32 == 0 && BN % 16 == 0 && BM % RPR == 0 && BREM % 64 == 0 && RPR % 16 == 0, "tile shape");
};

// ABL: timing-only ablations for tools/gemm_bench.py (wrong results): 1 = no global loads in the loop, 2 = no global
// loads and no ds_writes, 3 = no barrier, 4 = all of them (LDS reads + MFMAs only), 5 = loads but no ds_writes
// ROWOFF (round 4; the 128 x 128 tile only -- it has the registers): the A rows are gathered by 64-bit ELEMENT offsets
// (GemmOperand::rowoff: the feature rows of a batch handed over as per-clip pointer tables live in several allocations),
// one 64-bit pointer per staged piece instead of a 32-bit offset; GemmOperand::aux = the column split of a grouped launch.
// GANG (round 6; the N = 512 launch of vr_fc, K = 12 544): `dp_per_wg` (= GM, re-used: there are no whole tiles in this mode)
// x `half` (= the N-tiles) workgroups with consecutive ids form a gang that walks ONE stream-K range over (group of GM
// M-panels, K-step) in lockstep, member j on panel j / half of the group and N-tile j % half -- the members read the same
// rows of A (those on one panel) and the same rows of W (those on one N-tile) at the same time, on one XCD (xcd_remap), so an
// activation panel crosses the fabric once instead of once per N-tile and a weight panel once per GM M-panels.  (Plain
// stream-K gives every workgroup its own range: the four N-tiles of a panel are then computed at unrelated K offsets and
// `V` was read 4 x, L2 hit rate 4 %: profiles/r5_f_gemm_traffic_account.json.)  g_sk / sk_base / sk_rem count GANGS.
template <class T, class Epi, int ABL = 0, bool ROWOFF = false, bool GANG = false>
__global__ void __launch_bounds__(T::NT, 2)
gemm16_kernel(GemmOperand A, GemmOperand B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg,
              int g_sk, int sk_base, int sk_rem, int half, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB, AV = T::AV, BV = T::BV;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // fragment read offsets (floats) of this lane inside a 16-row block: row fr, slot (4 kb + fg) ^ ((fr >> 1) & 7)
  const int swz = (fr >> 1) & 7;
  const int frag0 = fr * kBK + ((fg ^ swz) << 2);          // kb = 0;  kb = 1 is frag0 ^ 16 (slot ^ 4)
  const int a_base = wave * 32 * kBK;                      // this wave's first A row
  // staging: thread = (row tid >> 3 of a 32-row round, 16-byte chunk tid & 7); LDS slot swizzled by the row
  const int srow = tid >> 3, schunk = tid & 7;
  const int st_off = srow * kBK + ((schunk ^ ((srow >> 1) & 7)) << 2);   // + round * RPR rows (RPR % 16 == 0: same swizzle)
  // B rows: BN = 176 is 5.5 rounds of 32 rows.  The last round has no branch: its upper half of the threads stages the
  // SAME rows as the lower half (row - 16: identical data to identical LDS addresses), so every thread runs the same
  // unconditional loads and stores
  constexpr int RPR = T::RPR;
  auto brow = [&](int i) { const int r = srow + RPR * i; return r < BN ? r : r - 16; };
  static_assert(T::BREM == 0 || BN - T::BVF * RPR >= 16, "partial B round: its idle rows re-stage the 16 rows in front of them");
  static_assert(T::BREM == 0 || T::BV * RPR - BN <= 16, "partial B round: at most 16 idle rows");
  const int st_last = st_off + (BV - 1) * RPR * kBK - (srow + RPR * (BV - 1) < BN ? 0 : 16 * kBK);

  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int gang_gm = GANG ? dp_per_wg : 1;                 // M-panels per gang (GANG re-uses the argument)
  if constexpr (GANG) dp_per_wg = 0;
  const int tiles_dp = dp_per_wg * G;
  const int gang_sz = GANG ? gang_gm * half : 1;
  const int gang_id = GANG ? blk / gang_sz : blk, gang_j = GANG ? blk - gang_id * gang_sz : 0;
  const int gang_n = GANG ? gang_j % half : 0, gang_m = GANG ? gang_j / half : 0;
  const SkRange rg = gang_id < g_sk ? sk_range(gang_id, sk_base, sk_rem) : SkRange{0, 0};
  // Every workgroup runs its whole tiles first and its stream-K range last, in step with its neighbours: the 64
  // workgroups of an XCD then work on one compact block of tiles at any moment and share its operand panels through that
  // XCD's L2.  (Round 3 tried to DE-synchronise them -- the stream-K range cut at the tile border it crosses and its
  // leading part run first, so that tile ends are spread over time instead of 512 workgroups storing 45 MB of C at once:
  // no faster, 140.3-141.0 vs 141.8-142.1 TFLOP/s, and 2.3 x the fabric reads, 3.3 vs 1.4 GB per [21120,1936,1936]
  // launch (rocprofv3 FETCH_SIZE), because workgroups that drift apart stop sharing panels.)
  const int pre_end = rg.begin;
#ifdef STTRAN_GEMM_EXPERIMENT
  if (g_t16_skew > 0) {
    const int units = ((blk & 7) + ((blk >> 3) & 7)) * g_t16_skew;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
  }
#endif

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
    if constexpr (GANG) {                                       // the range runs over groups of GM panels; this member's tile
      tile_m = tile * gang_gm + gang_m; tile_n = gang_n;
      if (tile_m >= tiles_m) { it += nsteps; continue; }      // (an odd panel count: the last group's upper members idle)
    } else tile_origin_rt(tile, tiles_m, tiles / tiles_m, half, tile_m, tile_n);     // `half` = N-tiles per group of the tile order
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#ifdef STTRAN_GEMM_EXPERIMENT
    unsigned long long clk0 = 0, clk1 = 0, clk2 = 0;
    if constexpr (ABL == 9) clk0 = __builtin_readcyclecounter();
#endif

    // global source of every staged piece: a uniform 64-bit base per TILE (the tile's first row of A / of W) plus a 32-bit
    // BYTE offset per lane -- one VGPR per piece instead of two, and no limit on the operand's size: an in-tile offset is
    // < (BM + 1) * ld * 4.  A GATHERED A operand (rowidx) has no tile-local base: its offsets are taken from A.ptr, and
    // gemm_linear only sends it here when its span is known to stay below 4 GB (GemmOperand::span).  Rows past M read
    // the tile's first row (gathered: row 0 of the operand): never stored by the epilogue.
    uint32_t oa[AV], ob[BV];
    const bool gathered = A.rowidx != nullptr;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int g = m0 + srow + RPR * i;
      const int64_t prow = g < M ? (gathered ? (int64_t)A.rowidx[g] : (int64_t)(g - m0)) : 0;
      oa[i] = (uint32_t)((prow * A.ld + schunk * 4 + ks0 * kBK) * 4);
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) ob[i] = (uint32_t)(((int64_t)brow(i) * B.ld + schunk * 4 + ks0 * kBK) * 4);
    const char* const abase = reinterpret_cast<const char*>(A.ptr) + (gathered ? (int64_t)0 : (int64_t)m0 * A.ld * 4);
    const char* pa64[ROWOFF ? AV : 1];
    if constexpr (ROWOFF) {
      const int64_t* ro = A.rowoff;
      if (A.aux > 0 && n0 >= A.aux) ro += M;                 // grouped launch: the second column group's gather table
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int g = m0 + srow + RPR * i;
        pa64[i] = reinterpret_cast<const char*>(A.ptr + ro[g < M ? g : m0] + schunk * 4 + ks0 * kBK);   // rows past M: the tile's first row
      }
    }
    (void)pa64;
    const char* const bbase = reinterpret_cast<const char*>(B.ptr) + (int64_t)n0 * B.ld * 4;
    // two register sets: the global loads of K-step u go to set u & 1, TWO steps ahead of their use (issued during step
    // u - 2, written to LDS during step u - 1), so a load has more than a whole K-step (~5 us) to arrive from HBM
    f32x4 ra[2][AV], rb[2][BV];
    auto koff = [&](int step) { return (step < nsteps ? step : 0) * kBK; };   // steps past the end re-read step 0 (never used)
    auto load_piece = [&](int set, int n, int ko) {
      if (n < AV) {
        if constexpr (ROWOFF) ra[set][n] = *reinterpret_cast<const f32x4*>(pa64[n] + (uint32_t)ko * 4u);
        else ra[set][n] = *reinterpret_cast<const f32x4*>(abase + (oa[n] + (uint32_t)ko * 4u));
      }
      else rb[set][n - AV] = *reinterpret_cast<const f32x4*>(bbase + (ob[n - AV] + (uint32_t)ko * 4u));
    };
    auto store_piece = [&](int set, int n, float* stage) {
      if (n < AV) *reinterpret_cast<f32x4*>(stage + st_off + n * RPR * kBK) = ra[set][n];
      else *reinterpret_cast<f32x4*>(stage + BM * kBK + (n - AV == BV - 1 ? st_last : st_off + (n - AV) * RPR * kBK)) = rb[set][n - AV];
    };

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NBLK = 2 * NB, NPL = AV + BV;
    static_assert(NPL <= NB, "staging pieces must fit the blocks of half a K-step");
#pragma unroll
    for (int n = 0; n < NPL; ++n) load_piece(0, n, koff(0));
#pragma unroll
    for (int n = 0; n < NPL; ++n) load_piece(1, n, koff(1));
#pragma unroll
    for (int n = 0; n < NPL; ++n) store_piece(0, n, smem);
    __syncthreads();

    // ---- software-pipelined main loop ------------------------------------------------------------------------------
    // A K-step is NBLK = 2 NB "blocks" (k16 group kb, column block j) of 8 MFMAs (2 row blocks x 4 k); block s uses the
    // B fragment fb[s & 1] and the A fragments fa[kb].  While block s runs, the fragment of block s + 1 is read from LDS
    // into the other register set (and the kb = 1 A fragments a few blocks ahead), one global load (of the K-step after
    // the next) rides in each of the first blocks and one ds_write (of the next K-step) in each of the last ones.  The
    // LAST block of a step is held in registers across the barrier: its MFMAs run after the first fragment reads of the
    // next step have been issued and cover the barrier wait and their latency (as the 32x32x2 kernel does with its last
    // group).  sched_group_barrier pins "MFMA, one memory instruction, MFMA ..." inside a block: left alone the compiler
    // issues every LDS read right in front of its first use and waits for it (a ~100-cycle stall every 8 MFMAs).
    f32x4 fa[2][2], fb[2];
    auto read_a = [&](const float* stage, int kb) {
      const int fo = kb ? (frag0 ^ 16) : frag0;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[kb][i] = *reinterpret_cast<const f32x4*>(stage + a_base + i * 16 * kBK + fo);
    };
    auto read_b = [&](const float* stage, int sblk) {
      const int kb = sblk / NB, j = sblk - kb * NB;
      fb[sblk & 1] = *reinterpret_cast<const f32x4*>(stage + (BM + j * 16) * kBK + (kb ? (frag0 ^ 16) : frag0));
    };
    auto mma_block = [&](int sblk) {
      const int kb = sblk / NB, j = sblk - kb * NB;
      // (e outer, i inner: two independent accumulators alternate, so an MFMA never waits for the 40-cycle
      //  dependent-accumulator latency of the one issued 32 cycles before it)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[sblk & 1][e], fa[kb][i][e], acc[i][j], 0, 0, 0);
    };
    read_a(smem, 0);
    read_b(smem, 0);
#ifdef STTRAN_GEMM_EXPERIMENT
    if constexpr (ABL == 9) clk1 = __builtin_readcyclecounter();
#endif
    // one K-step; `set` (compile-time) = t & 1: receives the loads of step t + 2, the other set (step t + 1) goes to LDS
    auto k_step = [&](int t, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const float* cur = smem + set * T::STAGE;
      float* nxt = smem + (set ^ 1) * T::STAGE;
      const int ko = koff(t + 2);
#pragma unroll
      for (int sb = 0; sb < NBLK; ++sb) {
        constexpr bool kLoads = ABL != 1 && ABL != 2 && ABL != 4, kStores = ABL != 2 && ABL != 4 && ABL != 5;
        if (sb + 1 < NBLK) read_b(cur, sb + 1);
        if (sb == NB - 3) read_a(cur, 1);                   // the kb = 1 row fragments, two blocks before their first use
        if (sb < NPL && kLoads) load_piece(set, sb, ko);
        if (sb >= NBLK - NPL && kStores) store_piece(set ^ 1, sb - (NBLK - NPL), nxt);
        if (sb + 1 < NBLK) {
          mma_block(sb);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (sb < NPL && kLoads) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          if (sb >= NBLK - NPL && kStores) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (sb == NB - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (sb == NB - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ABL != 3 && ABL != 4) __syncthreads();
      read_a(nxt, 0);
      read_b(nxt, 0);
      __builtin_amdgcn_sched_barrier(0);                    // issue these reads BEFORE the held-over block, which then hides them
      mma_block(NBLK - 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    // K = 32 k + 16 (1936 = 60.5 K-steps): the second half of the tile's last K-step multiplies by W's zero padding --
    // that step runs its kb = 0 blocks only (half a step of 61: 0.8 % of the MFMAs of every K = 1936 launch)
    auto k_half = [&](auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const float* cur = smem + set * T::STAGE;
#pragma unroll
      for (int sb = 0; sb < NB; ++sb) {
        if (sb + 1 < NB) {
          read_b(cur, sb + 1);
          mma_block(sb);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();                                      // every wave has read this stage: the next item may overwrite it
      mma_block(NB - 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      const bool half_tail = ABL == 0 && (K & 31) == 16 && ks1 == ksteps;
      const int nfull = half_tail ? nsteps - 1 : nsteps;
      int t = 0;
      for (; t + 1 < nfull; t += 2) {
        k_step(t, std::integral_constant<int, 0>{});
        k_step(t + 1, std::integral_constant<int, 1>{});
      }
      if (t < nfull) { k_step(t, std::integral_constant<int, 0>{}); ++t; }
      if (half_tail) {
        if (t & 1) k_half(std::integral_constant<int, 1>{}); else k_half(std::integral_constant<int, 0>{});
      }
    }
    if constexpr (ABL == 5) {          // loads issued but never written to LDS: keep them alive, wait for them only here
#pragma unroll
      for (int n = 0; n < AV; ++n) asm volatile("" ::"v"(ra[0][n]), "v"(ra[1][n]));
#pragma unroll
      for (int n = 0; n < BV; ++n) asm volatile("" ::"v"(rb[0][n]), "v"(rb[1][n]));
    }

#ifdef STTRAN_GEMM_EXPERIMENT
    if constexpr (ABL == 9) clk2 = __builtin_readcyclecounter();
#endif
    // C layout (weights on the "A" port): output row = lane % 16 of block i, columns 4 (lane / 16) + {0..3} of block j
    const int row0 = m0 + wave * 32 + fr;
    const int col0 = n0 + 4 * fg;
    if (nsteps == ksteps) {
      // One strip per output row: ALL of the row's loads (bias, position bias, residual) are issued before its first
      // store.  vmcnt counts stores too on gfx9 and memory operations retire in order, so a load issued behind stores
      // waits for their acknowledgement: every load -> store phase costs a load latency plus a store round trip.  The
      // 176-column tile has no registers for a whole row's operands next to its 88 accumulator registers (it spills):
      // its rows go in two halves (measured: 140.2 vs 138.2 TFLOP/s on [21120,1936,1936] + residual; the 128-column
      // tile gains 1 % from whole rows)
      const int rows2[2] = {row0, row0 + 16};
      const bool valid2[2] = {row0 < M, row0 + 16 < M};
      // the common forms (bias + residual, bias + position bias, bias only): both rows in one load -> store phase (the
      // 128-column tile), or in two -- columns [0, 96) and [96, 176) -- on the 176-column tile, whose 88 accumulator
      // registers leave no room for a whole row pair's operands (33 vectors: the kernel spilled 24 registers around them)
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
    } else {
      // partial K range: park the raw accumulators, register-major (1 KB per wave-instruction); gemm16_fixup_kernel sums
      // them in ascending workgroup order (deterministic) and runs the epilogue
      f32x4* sp = reinterpret_cast<f32x4*>(slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN)) + tid;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) sp[(i * NB + j) * NT] = acc[i][j];
    }
#ifdef STTRAN_GEMM_EXPERIMENT
    if constexpr (ABL == 9) {
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long clk3 = __builtin_readcyclecounter();
      if (tid == 0) {
        atomicAdd(&g_t16_clk[0], clk1 - clk0); atomicAdd(&g_t16_clk[1], clk2 - clk1); atomicAdd(&g_t16_clk[2], clk3 - clk2);
        atomicAdd(&g_t16_clk[3], 1ull); atomicAdd(&g_t16_clk[4], (unsigned long long)nsteps);
        if (g_t16_trace) {
          const unsigned int r = atomicAdd(&g_t16_trace_n, 1u);
          if (r < g_t16_trace_cap) {
            unsigned long long* rec = g_t16_trace + (size_t)r * 8;
            rec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
            rec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
            rec[2] = (unsigned long long)blk; rec[3] = ((unsigned long long)nsteps << 32) | (unsigned int)tile;
            rec[4] = clk0; rec[5] = clk1; rec[6] = clk2; rec[7] = clk3;
          }
        }
      }
    }
#endif
    if (!dp) it += nsteps;
  }
}

// grid = (stream-K tiles, 2 * NB): one workgroup per accumulator register group of a split tile
// GANG: blockIdx.x = panel * group_n + N-tile, tiles_dp = GM (panels per gang; re-used); the owners are gangs: gang b's member
// for (panel % GM, N-tile) parked at slab block b * GM * group_n + (panel % GM) * group_n + N-tile
template <class T, class Epi, bool GANG = false>
__global__ void __launch_bounds__(T::NT)
gemm16_fixup_kernel(int M, int N, int tiles_m, int tiles_n, int ksteps, int g_sk, int sk_base, int sk_rem, int tiles_dp,
                    int group_n, const float* __restrict__ slab, Epi epi) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB;
  const int gang_gm = GANG ? tiles_dp : 1;
  if constexpr (GANG) tiles_dp = 0;
  const int gang_n = GANG ? (int)blockIdx.x % group_n : 0;
  const int panel = GANG ? (int)blockIdx.x / group_n : 0;
  const int tile = GANG ? panel / gang_gm : (int)blockIdx.x;          // GANG: the group of panels the owners' ranges run over
  const int gang_j = GANG ? (panel - tile * gang_gm) * group_n + gang_n : 0;
  const int t0 = tile * ksteps, t1 = t0 + ksteps;
  const int b_lo = sk_owner(t0, sk_base, sk_rem), b_hi = sk_owner(t1 - 1, sk_base, sk_rem);
  if (b_lo == b_hi) return;                      // computed whole by one workgroup: nothing parked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int ij = blockIdx.y, i = ij / NB, j = ij % NB;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* base = reinterpret_cast<const f32x4*>(slab) + (int64_t)ij * NT + tid;
  for (int b = b_lo; b <= b_hi; b += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = b + u;
      const bool ok = bb <= b_hi;
      const int slot = (bb == b_lo && sk_range(bb, sk_base, sk_rem).begin < t0) ? 1 : 0;
      const int64_t sb = GANG ? (int64_t)(ok ? bb : b_lo) * (gang_gm * group_n) + gang_j : (int64_t)(ok ? bb : b_lo);
      const f32x4* sp = base + (sb * 2 + slot) * (BM * BN / 4);
      v[u] = ok ? *sp : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const int gt = tiles_dp + tile;
  int tile_m, tile_n;
  if constexpr (GANG) { tile_m = panel; tile_n = gang_n; }
  else tile_origin_rt(gt, tiles_m, tiles_n, group_n, tile_m, tile_n);
  const int row = tile_m * BM + wave * 32 + 16 * i + fr;
  const int col = tile_n * BN + 16 * j + 4 * fg;
  if (row < M) epi.vec(row, col, acc);
}

}  // namespace sttran
