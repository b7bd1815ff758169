// gemm_f32_t16c.h -- the two convolutions of the pair fusion on the 16x16x4 kernel structure of gemm_f32_t16.h.
//
//   C[ch, n] = W[ch, :] . X[n, :]     ch = 256 output channels (ALL of them in one 256 x 128 tile, so the big operand is
//                                     fetched once), n = pair * 49 + hw running straight over the pair borders
//   B_UNION_FLAT  union_func1 (lib/sttran.py:336,386): X[n, k] = union_feat[pair][k][hw], read in place from the NCHW
//                 tensor -- a thread stages (column, 4 consecutive k) with four dword loads (the lanes of a wave walk 64
//                 consecutive columns) and ONE ds_write_b128: the transposition happens in the registers;
//                 V[pair][ch][hw] += acc + bias[ch]: the accumulators START from V, the epilogue is a plain store
//   B_CONV2       Conv2d(128, 256, 3, padding 1) of the mask branch (lib/sttran.py:342) as an implicit GEMM: K ordered
//                 (ky, kx, ci) over the channel-last pooled map C2[pair][7][7][128], one tap per four K-steps, a piece =
//                 4 consecutive channels of one tap = one 16-byte load, taps outside the image are zero pieces;
//                 V[pair][ch][hw] = BN(ReLU(acc + bias))   (ReLU before BN: lib/sttran.py:342-344)
//
// Same machinery as the nn.Linear tile: unpadded 128-byte LDS rows with the 16-byte slot XOR-ed by (row >> 1) & 7
// (conflict-free ds_read_b128 of 16-row fragments), one ds_read_b128 per four MFMAs, global loads two K-steps ahead in
// two register sets, one memory instruction per MFMA gap, the last block of a K-step held across the barrier, hybrid
// data-parallel + stream-K schedule with parked partials and a fix-up launch.  Differences: 8 waves (one workgroup per
// CU: 96 KB of LDS), the weights sit on the MFMA's "A" port so that a lane holds FOUR CHANNELS of ONE column -- the 16
// lanes of a group then store 16 consecutive hw of a channel (64 contiguous bytes of V) --, and the per-channel epilogue
// constants are loaded before the first store (no load waits behind a store that might alias it).
#pragma once
#include <type_traits>

#include "gemm_f32_mfma.h"

namespace sttran {

// four floats at a 4-byte aligned address: global memory takes a dwordx4 access at any dword address (unaligned access mode)
struct __attribute__((packed, aligned(4))) V4a4 { f32x4 v; };

template <int BKIND_>
struct Tile16C {
  static constexpr int BM = 256, BN = 128, BKIND = BKIND_;
  static constexpr int WAVES = BM / 32, NT = WAVES * 64;   // 8 waves, wave w owns channels 32 w .. 32 w + 31
  static constexpr int NB = BN / 16;
  static constexpr int STAGE = (BM + BN) * kBK;
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int RPR = NT / 8;                       // 64 rows per staging round (8 threads per 128-byte row)
  static constexpr int AV = BM / RPR;                      // 4 weight pieces per thread and K-step
  static constexpr int BV = BN * 8 / NT;                   // 2 column pieces
  static constexpr int GROUP_N = 8;
  static_assert(BKIND == B_UNION_FLAT || BKIND == B_CONV2, "convolution operand kinds only");
};

// ABL (experiment builds, STTRAN_T16C_ABLATE; timing only, wrong results): 1 = no column-operand (B) global loads in the
// loop, 2 = no weight (A) loads, 3 = no loads at all, 4 = no loads and no ds_writes, 5 = no barrier, 6 = accumulators start
// from zero (no V read at the tile start), 7 = no epilogue stores, 8 = the B loads are issued but nothing waits for them
// (they land in registers the ds_writes do not read): separates their issue / bandwidth cost from their latency

// One K range [ks0, ks0 + nsteps) of the tile at columns n0 .. n0 + 127 (all 256 channels: m0 = 0) accumulated into `acc`.
// Shared by gemm16c_kernel (one convolution per launch) and pair_conv_fused_kernel (conv3x3, its ReLU / BN, then the union
// conv's K range on the same accumulators).  On return every wave has passed the last barrier of the loop and reads the stage
// buffers no more (what it still reads there is the unused look-ahead of a step that does not exist).
template <class T, int ABL>
__device__ __forceinline__ void conv_kloop(const GemmOperand& A, const GemmOperand& B, int N, int m0, int n0, int ks0, int nsteps,
                                           float* smem, f32x4 (&acc)[2][T::NB]) {
  constexpr int BM = T::BM, BN = T::BN, NB = T::NB, AV = T::AV, BV = T::BV, RPR = T::RPR;
  constexpr bool UFLAT = T::BKIND == B_UNION_FLAT;
  using Geo = ConvGeo<B_CONV2>;
  // the thread id through an opaque move: the index arithmetic below is then recomputed per call and dies with it (the fused
  // kernel calls this twice per tile with different operand kinds: with everything derived from ONE threadIdx.x the compiler
  // kept both phases' staging offsets and pointers alive across the whole tile loop and spilled)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = (fr >> 1) & 7;
  const int frag0 = fr * kBK + ((fg ^ swz) << 2);          // kb = 0; kb = 1 is frag0 ^ 16
  const int a_base = wave * 32 * kBK;
  const int srow = tid >> 3, schunk = tid & 7;
  const int st_a = srow * kBK + ((schunk ^ ((srow >> 1) & 7)) << 2);     // weight piece i: + i * RPR rows
  // column pieces.  CONV2: (row srow + RPR i, chunk schunk) like the weights.  UNION: thread = (column tid % BN, chunks
  // 2 (tid / BN) + i): consecutive lanes are consecutive columns = consecutive hw of one channel row in memory
  const int ucol = tid % BN, uch0 = (tid / BN) * BV;
  int st_b[BV];
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    const int r = UFLAT ? ucol : srow + RPR * i, ch = UFLAT ? uch0 + i : schunk;
    st_b[i] = (BM + r) * kBK + ((ch ^ ((r >> 1) & 7)) << 2);
  }
  // ---- sources of the staged pieces ---------------------------------------------------------------------------------
  uint32_t oa[AV];                                           // weights: M = 256 rows exactly, K % 32 == 0 (launcher checks)
#pragma unroll
  for (int i = 0; i < AV; ++i) oa[i] = (uint32_t)((((int64_t)(m0 + srow + RPR * i)) * A.ld + schunk * 4 + ks0 * kBK) * 4);
  const char* const abase = reinterpret_cast<const char*>(A.ptr);
  const float* pb[BV];                                       // columns: a pointer per piece (the tensors exceed 4 GB)
  int cy[BV], cx[BV];                                        // CONV2: input row / column of tap (0, 0), -100 = column past N
  (void)cy; (void)cx;
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    const int n = n0 + (UFLAT ? ucol : srow + RPR * i);
    const bool v = n < N;
    const int nn = v ? n : 0, p = nn / kUHW, hw = nn - p * kUHW;
    if constexpr (UFLAT) {
      pb[i] = B.ptr + (B.rowoff ? B.rowoff[p] : (int64_t)p * B.ld) + hw + (int64_t)(uch0 + i) * 4 * kUHW;
    } else {
      const int oy = hw / Geo::HO, ox = hw - oy * Geo::HO;
      cy[i] = v ? oy - Geo::PAD : -100; cx[i] = ox - Geo::PAD;
      pb[i] = B.ptr + (int64_t)p * (Geo::CIN * Geo::HI * Geo::HI) + schunk * 4;
    }
  }
  f32x4 ra[2][AV], rb[2][BV];
  f32x4 rbx[2][BV];                                          // ABL 8 only
  (void)rbx;
  bool zb[2][BV];                                            // CONV2: the piece is a zero piece (tap outside the image)
  (void)zb;
  auto kstep_of = [&](int step) { return ks0 + (step < nsteps ? step : 0); };   // steps past the end re-read step 0 (unused)
  // load slot n of a K-step: AV weight pieces, then the column pieces (a UNION piece is four dword loads, each its own slot)
  // LOAD order (round 4, union conv only): the column operand first, the weights last -- union_feat comes from HBM, the
  // weights are 2 MB that every tile re-reads from L2; the column pieces get a quarter of a K-step more time in flight:
  // 4 246 vs 4 270 us in situ (conv3x3, whose gathered operand is cache-resident, lost 0.6 % with it and keeps the old
  // order).  Slot m of the K-step loads piece n = (m + AV) mod NLS.
  constexpr int NLS = AV + (UFLAT ? 4 * BV : BV);
  auto load_slot = [&](int set, int m, int step) {
    const int ks = kstep_of(step);
    const int n = !UFLAT ? m : (m < NLS - AV ? m + AV : m - (NLS - AV));
    if (n < AV) {
      ra[set][n] = *reinterpret_cast<const f32x4*>(abase + (oa[n] + (uint32_t)((ks - ks0) * kBK * 4)));
    } else if constexpr (UFLAT) {
      const int i = (n - AV) >> 2, e = (n - AV) & 3;
      if constexpr (ABL == 8) rbx[set][i][e] = pb[i][((int64_t)ks * kBK + e) * kUHW];
      else rb[set][i][e] = pb[i][((int64_t)ks * kBK + e) * kUHW];
    } else {
      const int i = n - AV;
      const int k0 = ks * kBK, tap = k0 / Geo::CIN, ky = tap / Geo::KH, kx = tap - ky * Geo::KH, ci0 = k0 - tap * Geo::CIN;
      const int iy = cy[i] + ky, ix = cx[i] + kx;
      const bool ok = (unsigned)iy < (unsigned)Geo::HI && (unsigned)ix < (unsigned)Geo::HI;
      if constexpr (ABL == 8) rbx[set][i] = *reinterpret_cast<const f32x4*>(pb[i] + (ok ? (iy * Geo::HI + ix) * Geo::CIN + ci0 : 0));
      else rb[set][i] = *reinterpret_cast<const f32x4*>(pb[i] + (ok ? (iy * Geo::HI + ix) * Geo::CIN + ci0 : 0));
      zb[set][i] = !ok;
    }
  };
  constexpr int NP = AV + BV;
  auto store_piece = [&](int set, int n, float* stage) {
    if (n < AV) *reinterpret_cast<f32x4*>(stage + st_a + n * RPR * kBK) = ra[set][n];
    else {
      const int i = n - AV;
      if constexpr (UFLAT) *reinterpret_cast<f32x4*>(stage + st_b[i]) = rb[set][i];
      else *reinterpret_cast<f32x4*>(stage + st_b[i]) = zb[set][i] ? f32x4{0.f, 0.f, 0.f, 0.f} : rb[set][i];
    }
  };

  constexpr int NBLK = 2 * NB;
  static_assert(NLS <= NBLK - 1 && NP <= NBLK - 1, "staging slots must fit the blocks of a K-step");
#pragma unroll
  for (int n = 0; n < NLS; ++n) load_slot(0, n, 0);
#pragma unroll
  for (int n = 0; n < NLS; ++n) load_slot(1, n, 1);
#pragma unroll
  for (int n = 0; n < NP; ++n) store_piece(0, n, smem);
  __syncthreads();

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
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kb][i][e], fb[sblk & 1][e], acc[i][j], 0, 0, 0);
  };
  read_a(smem, 0);
  read_b(smem, 0);
  auto k_step = [&](int t, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const float* cur = smem + set * T::STAGE;
    float* nxt = smem + (set ^ 1) * T::STAGE;
#pragma unroll
    for (int sb = 0; sb < NBLK; ++sb) {
      if (sb + 1 < NBLK) read_b(cur, sb + 1);
      if (sb == NB - 3) read_a(cur, 1);
      constexpr bool kLoadB = ABL != 1 && ABL != 3 && ABL != 4, kLoadA = ABL != 2 && ABL != 3 && ABL != 4, kStore = ABL != 4;
      if (sb < NLS && ((UFLAT ? sb >= NLS - AV : sb < AV) ? kLoadA : kLoadB)) load_slot(set, sb, t + 2);
      if (sb >= NBLK - NP && kStore) store_piece(set ^ 1, sb - (NBLK - NP), nxt);
      if (sb + 1 < NBLK) {
        mma_block(sb);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (sb < NLS && ((UFLAT ? sb >= NLS - AV : sb < AV) ? kLoadA : kLoadB)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (sb >= NBLK - NP && kStore) {
          if constexpr (!UFLAT) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (sb == NB - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (sb == NB - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ABL != 5) __syncthreads();
    read_a(nxt, 0);
    read_b(nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_block(NBLK - 1);
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

  if constexpr (ABL == 8) {
#pragma unroll
    for (int i = 0; i < BV; ++i) asm volatile("" ::"v"(rbx[0][i]), "v"(rbx[1][i]));
  }
}

// The finished tile (all of its K range in `acc`) through the epilogue: value() per element, then out through the wave's LDS block
template <class T, class Epi, int ABL>
__device__ __forceinline__ void conv_store_tile(const Epi& epi, int N, int m0, int n0, float* smem, f32x4 (&acc)[2][T::NB]) {
  constexpr int BN = T::BN, NB = T::NB;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                      // (see conv_kloop)
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ch0 = m0 + wave * 32 + 4 * fg;
  // Whole tile.  Round 3-4 stored straight from the accumulators: a lane holds four CHANNELS of one column, so a store
  // instruction wrote 4 x 64 contiguous bytes (16 hw of 4 channels) and a tile took 64 dword stores per lane (2.8-4.3 % of
  // the kernel by ablation).  Now the finished values cross the wave's own LDS block (16 channels x 128 columns per pass,
  // rows 132 floats apart) and leave as 16-byte pieces of four consecutive hw: a [49]-float row of V starts at any
  // multiple of 4 bytes, which global memory takes for a dwordx4 access.  A piece that would run over a pair's last hw
  // (the tile's columns run straight over the pair borders: at most three such pieces in 128 columns) or over column N
  // is left out of the 16-byte pass; those few pieces x 16 channels are spread over the lanes of a second, dword pass.
  constexpr int EPS = BN + 4;                        // floats per LDS row: 528 bytes = 33 sixteen-byte slots
  static_assert(8 * 16 * EPS * 4 <= T::LDS_BYTES, "eight wave blocks fit the stage buffers");
  typename Epi::Consts cst[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) cst[i][v] = epi.consts(ch0 + 16 * i + v);
  // this lane's piece of every row it handles: columns n0 + 4 q .. + 3
  const int q = lane & 31, c0 = n0 + 4 * q;
  const int pq = c0 / kUHW, hwq = c0 - pq * kUHW;
  const bool whole = hwq + 3 < kUHW && c0 + 3 < N;  // inside one pair, inside the operand
  float* const vq = whole ? epi.addr(m0 + wave * 32 + (lane >> 5), c0) : nullptr;
  // the pieces left to the dword pass: the one before every pair border inside the tile (unless the border falls on a
  // piece border) and the one that holds column N
  int odd[4], nodd = 0;
  {
    const int b0 = (n0 + kUHW - 1) / kUHW * kUHW;   // first pair border >= n0
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b = b0 + k * kUHW, r = b - n0;
      if (r < BN && b < N && (r & 3)) odd[nodd++] = r >> 2;
    }
    if (N - n0 < BN && N > n0 && ((N - n0) & 3)) odd[nodd++] = (N - n0) >> 2;
  }
  __syncthreads();                                   // every wave is done with the stage buffers
  float* const ep = smem + wave * (16 * EPS);
  const bool keep = ABL != 7 || acc[0][0][0] == 12345.678f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) ep[(4 * fg + v) * EPS + 16 * j + fr] = epi.value(acc[i][j][v], cst[i][v]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the block is private to this wave: program order is enough
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (whole && keep) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {                  // rows (lane >> 5) + 2 u of the pass, 16 bytes each
        const f32x4 val = *reinterpret_cast<const f32x4*>(ep + ((lane >> 5) + 2 * u) * EPS + 4 * q);
        reinterpret_cast<V4a4*>(vq + (int64_t)(16 * i + 2 * u) * kUHW)->v = val;
      }
    }
    if (nodd && keep) {                              // (item = odd piece x row) per lane, four dword stores each
      for (int it = lane; it < nodd * 16; it += 64) {
        const int k = it >> 4, r = it & 15, qq = k == 0 ? odd[0] : k == 1 ? odd[1] : k == 2 ? odd[2] : odd[3];
        const f32x4 val = *reinterpret_cast<const f32x4*>(ep + r * EPS + 4 * qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = n0 + 4 * qq + e;
          if (col < N) *epi.addr(m0 + wave * 32 + 16 * i + r, col) = val[e];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next pass's writes stay behind these reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();                                   // the next tile's first K-step is staged over these blocks
}

template <class T, class Epi, int ABL = 0>
__global__ void __launch_bounds__(T::NT, 2)
gemm16c_kernel(GemmOperand A, GemmOperand B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg,
               int g_sk, int sk_base, int sk_rem, int tile_base, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int tiles_dp = dp_per_wg * G;
  const SkRange rg = blk < g_sk ? sk_range(blk, sk_base, sk_rem) : SkRange{0, 0};

  int dp_done = 0;
  for (int it = rg.begin; dp_done < dp_per_wg || it < rg.end;) {
    int tile, ks0, ks1;
    const bool dp = dp_done < dp_per_wg;
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
    tile += tile_base;                                         // a launch over the tiles behind pair_conv_fused_kernel's
    const int nsteps = ks1 - ks0;
    int tile_m, tile_n;
    tile_origin<T::GROUP_N>(tile, tiles_m, tiles / tiles_m, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // accumulators: lane (fr, fg) holds, for row block i and column block j, channels 32 w + 16 i + 4 fg + {0..3} of
    // column n0 + 16 j + fr
    f32x4 acc[2][NB];
    const int ch0 = m0 + wave * 32 + 4 * fg;
    const int colb = n0 + fr;
    if constexpr (EpiInit<Epi>::value && ABL != 6) {
      if (ks0 == 0) {
        // C += A B: the K range that starts a tile accumulates onto the output's old values (64 independent loads,
        // in flight with the first operand loads)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int col = colb + 16 * j;
              acc[i][j][v] = col < N ? epi.init(ch0 + 16 * i + v, col) : 0.f;
            }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    conv_kloop<T, ABL>(A, B, N, m0, n0, ks0, nsteps, smem, acc);
    if (nsteps == ksteps) {
      conv_store_tile<T, Epi, ABL>(epi, N, m0, n0, smem, acc);
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

// grid = (stream-K tiles, 2 * NB): sums the parked partial accumulators of a split tile in ascending workgroup order
template <class T, class Epi>
__global__ void __launch_bounds__(T::NT)
gemm16c_fixup_kernel(int M, int N, int tiles_m, int tiles_n, int ksteps, int g_sk, int sk_base, int sk_rem, int tiles_dp,
                     const float* __restrict__ slab, Epi epi) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NB = T::NB;
  const int tile = blockIdx.x;
  const int t0 = tile * ksteps, t1 = t0 + ksteps;
  const int b_lo = sk_owner(t0, sk_base, sk_rem), b_hi = sk_owner(t1 - 1, sk_base, sk_rem);
  if (b_lo == b_hi) return;
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
      const f32x4* sp = base + ((int64_t)(ok ? bb : b_lo) * 2 + slot) * (BM * BN / 4);
      v[u] = ok ? *sp : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const int gt = tiles_dp + tile;
  int tile_m, tile_n;
  tile_origin<T::GROUP_N>(gt, tiles_m, tiles_n, tile_m, tile_n);
  const int ch = tile_m * BM + wave * 32 + 16 * i + 4 * fg;
  const int col = tile_n * BN + 16 * j + fr;
  if (col < N) {
#pragma unroll
    for (int v = 0; v < 4; ++v) epi.store(ch + v, col, acc[v], epi.consts(ch + v));
  }
}

// ---- epilogues: row = output channel, col = pair * 49 + hw ----------------------------------------------------------
// union_func1: the partial sums START from V (EpiInit), so the tile's epilogue -- or the fix-up's -- adds the bias and stores
struct EpiUnionT16 {
  float* V; const float* bias; int C;
  static constexpr bool kInit = true;
  struct Consts { float b; };
  __device__ __forceinline__ float* at(int row, int col) const {
    const int p = col / kUHW, hw = col - p * kUHW;
    return V + ((int64_t)p * C + row) * kUHW + hw;
  }
  __device__ __forceinline__ float init(int row, int col) const { return *at(row, col); }
  __device__ __forceinline__ Consts consts(int row) const { return Consts{bias[row]}; }
  __device__ __forceinline__ float value(float v, const Consts& c) const { return v + c.b; }
  __device__ __forceinline__ float* addr(int row, int col) const { return at(row, col); }
  __device__ __forceinline__ void store(int row, int col, float v, const Consts& c) const { *at(row, col) = value(v, c); }
};
// conv3x3: ReLU, then eval-mode BatchNorm (lib/sttran.py:342-344), channel-major into V[p][c][hw]
struct EpiConvT16 {
  float* V; const float* bias; const float* scale; const float* shift; int C;
  struct Consts { float b, s, t; };
  __device__ __forceinline__ Consts consts(int row) const { return Consts{bias[row], scale[row], shift[row]}; }
  __device__ __forceinline__ float value(float v, const Consts& c) const { return relu_nan(v + c.b) * c.s + c.t; }
  __device__ __forceinline__ float* addr(int row, int col) const {
    const int p = col / kUHW, hw = col - p * kUHW;
    return V + ((int64_t)p * C + row) * kUHW + hw;
  }
  __device__ __forceinline__ void store(int row, int col, float v, const Consts& c) const { *addr(row, col) = value(v, c); }
};

// ---- both convolutions of a pair block in ONE pass over the tile (round 6) ------------------------------------------------------
// V[p][ch][hw] = BN(ReLU(conv3x3(C2) + b4)) + union_func1(U) + b1   (lib/sttran.py:342-345 + :386-387, summed at :388).
// As two launches the conv3x3 stores its 256 x 128 tile (64 store instructions per lane) and the union conv reads it back as
// its accumulators' start values (64 loads per lane): 2 x 565 MB at 64 clips of 16x12, one pipeline fill and drain per tile
// and launch, one fix-up launch.  Here a workgroup runs the conv3x3's 36 K-steps, applies bias / ReLU / BN to the
// accumulators in registers, runs the union conv's K-steps ON THEM and stores once -- the same arithmetic in the same order
// (bit-identical to the two launches on tiles neither of them splits).  The ReLU between the two K ranges makes a tile
// indivisible for stream-K, so this kernel takes whole tiles only: the launch's whole rounds, plus the leftover tiles
// when they fill most of another round (pair_convs_fused_tiles); otherwise the leftover goes through the two single-convolution
// launches with `tile_base` (their stream-K balances the tail).
template <int ABL = 0>
__global__ void __launch_bounds__(Tile16C<B_CONV2>::NT, 2)
pair_conv_fused_kernel(GemmOperand A2, GemmOperand B2, GemmOperand A1, GemmOperand B1, int N, int K1, int ntiles, EpiConvT16 e2,
                       EpiUnionT16 e1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using T2 = Tile16C<B_CONV2>;
  using T1 = Tile16C<B_UNION_FLAT>;
  constexpr int NB = T2::NB;
  static_assert(T1::LDS_BYTES == T2::LDS_BYTES && T1::NT == T2::NT && T1::BN == T2::BN, "one tile geometry");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fg = lane >> 4;
  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int ch0 = wave * 32 + 4 * fg;
  for (int tile = blk; tile < ntiles; tile += G) {           // whole tiles only; the last round may be partly filled
    const int n0 = tile * T2::BN;
    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv_kloop<T2, ABL>(A2, B2, N, 0, n0, 0, ConvGeo<B_CONV2>::KREAL / kBK, smem, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const EpiConvT16::Consts c = e2.consts(ch0 + 16 * i + v);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j][v] = e2.value(acc[i][j][v], c);
      }
    conv_kloop<T1, ABL>(A1, B1, N, 0, n0, 0, K1 / kBK, smem, acc);
    conv_store_tile<T1, EpiUnionT16, ABL>(e1, N, 0, n0, smem, acc);
  }
}

}  // namespace sttran
