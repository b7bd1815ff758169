// gemm_bf16x3.h -- fp32 GEMM emulated on the bf16 matrix pipe (EXPERIMENT: opt-in, not the product's default path).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Every fp32 operand x is split into three bf16 planes x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 3 x 8 mantissa bits = the 24 of fp32), and a product a.b is evaluated as the six cross terms
// a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  A bf16 x bf16 product is
// exact in fp32, and the three dropped terms (a2b3, a3b2, a3b3) are below 2^-24 |a||b|: the result carries fp32-level
// error, at 16 / 6 = 2.7x the rate of v_mfma_f32_32x32x2_f32 (ceiling 419 TFLOP/s-equivalent on MI355X).
//
// Weights arrive already split (three bf16 planes [3][N][ldp], zero-padded to a multiple of 32 columns, made once at
// load time); activations stay fp32 in HBM and are split by the A loader on their way into LDS (v_cvt_pk_bf16_f32 +
// shift + subtract: ~5 VALU operations per element, issued under the MFMAs).
//
// Layout: per LDS stage three planes of [BM + BN rows][32 k] bf16 (64 bytes per row, unpadded); the ds_read_b128
// fragment reads (lane = row & 31, k half = lane >> 5: 8 consecutive k per lane, what the 32x32x16 MFMA wants) are kept
// conflict-free by XOR-ing the 16-byte chunk index with (row >> 2) & 3.  One fragment read feeds three MFMAs (each
// plane of A meets two or three planes of B), i.e. half the LDS traffic per MFMA of a plain bf16 GEMM.
// Swapped operand ports and the 16-byte vector epilogue, the hybrid data-parallel + stream-K schedule and the fix-up
// launch are those of gemm_f32_mfma.h (same accumulator layout, same parking format).
#pragma once
#include "gemm_f32_mfma.h"

namespace sttran {

#ifndef X3_ABLATE
#define X3_ABLATE 0          // timing-only ablations (wrong results): 1 = no split arithmetic, 2 = no staging at all in the loop
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct X3Weights {
  const __bf16* planes;     // [3][rows][ldp]
  int64_t ldp;              // row stride in elements (multiple of 32)
  int64_t plane_stride;     // rows * ldp
};

// A-operand kinds.  A_ROWS: K-contiguous fp32 rows (every nn.Linear input).  A_UNION_FLAT: the NCHW union_feat tensor
// U[P][K][49] read in place as the ACTIVATION operand of union_func1 -- GEMM row = pair * 49 + hw, running over the pair
// borders like B_UNION_FLAT of gemm_f32_mfma.h (there the tensor is the B operand; here the weights are, because they
// are the side that arrives pre-split).  A thread stages (row, 4 consecutive k): four coalesced dword loads (the lanes of
// a wave walk consecutive rows = hw), split in registers, one 8-byte write per plane.
// A_CONV2: the 3x3 convolution of the mask branch (lib/sttran.py:342) as an implicit GEMM with the activations as the A
// operand: GEMM row = (pair, oy, ox), column k = (ky, kx, ci) gathered from the channel-last input [pair][7][7][128] on
// the fly (one tap = 128 channels = 4 K-steps; a tap outside the image is a zero piece).
enum { A_ROWS = 0, A_UNION_FLAT = 1, A_CONV2 = 2 };

template <int BM_, int BN_, int WM_, int WN_, int AKIND_ = A_ROWS>
struct X3Tile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, AKIND = AKIND_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int ROWS = BM + BN;
  static constexpr int AV = BM * 8 / NT;                    // float4 loads per thread per K-step (A, fp32)
  static constexpr int BV = 3 * BN * 4 / NT;                // 16-byte loads per thread per K-step (B, three bf16 planes)
  static constexpr int STAGE_BYTES = 3 * ROWS * 64;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int GROUP_N = 8;
  static_assert((BM * 8) % NT == 0 && (3 * BN * 4) % NT == 0, "staging must divide evenly");
  static_assert(AKIND != A_UNION_FLAT || (NT % BM == 0 && (NT / BM) * AV == 8), "A_UNION_FLAT: (row, k-group) per thread");
};

// union_func1 with the roles of the bf16x3 engine: row = pair * 49 + hw, col = out channel; `V += acc + bias` as
// "accumulate onto V" (kInit, see EpiUnionFlat): V[p][c][hw] is read into the accumulators of the K range that starts a
// tile and stored by the epilogue.  Lanes hold consecutive rows = consecutive hw of one channel: 128-byte runs.
// conv3x3 -> ReLU -> eval-BatchNorm with the engine's roles (row = (pair, position), col = out channel), stored
// channel-major into V[p][c][49] like EpiConvRelBn
struct EpiConvRows {
  float* V; const float* bias; const float* scale; const float* shift; int C;
  static constexpr bool kVector = true;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    const int p = row / kUHW, pos = row - p * kUHW;
    V[((int64_t)p * C + col) * kUHW + pos] = relu_nan(v + bias[col]) * scale[col] + shift[col];
  }
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) (*this)(row, col + c, v[c]);
  }
};
struct EpiUnionRows {
  float* V; const float* bias; int C;
  static constexpr bool kVector = true;       // swapped MFMA ports (the engine's only form); vec() is four scalar stores
  static constexpr bool kInit = true;
  __device__ __forceinline__ float* at(int row, int col) const {
    const int p = row / kUHW, hw = row - p * kUHW;
    return V + ((int64_t)p * C + col) * kUHW + hw;
  }
  __device__ __forceinline__ bool init_on() const { return true; }
  __device__ __forceinline__ float init(int row, int col) const { return *at(row, col); }
  __device__ __forceinline__ void operator()(int row, int col, float v) const { *at(row, col) = v + bias[col]; }
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) (*this)(row, col + c, v[c]);
  }
};

// x -> (hi, mid, lo) bf16 planes, four elements at a time
__device__ __forceinline__ void split3(const f32x4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (__bf16)v[i];
    const float r1 = v[i] - (float)h[i];
    m[i] = (__bf16)r1;
    l[i] = (__bf16)(r1 - (float)m[i]);
  }
}

// One-off: split a [rows, cols] fp32 matrix (row stride ld) into three zero-padded bf16 planes [3][rows][ldp].
static __global__ void __launch_bounds__(256)
split_planes_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols, __bf16* __restrict__ planes, int64_t ldp) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < (int)ldp; c += 256) {
    const float v = c < cols ? src[(int64_t)r * ld + c] : 0.f;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const int64_t o = (int64_t)r * ldp + c, ps = (int64_t)rows * ldp;
    planes[o] = h; planes[ps + o] = m; planes[2 * ps + o] = l;
  }
}

template <class T, class Epi>
__global__ void __launch_bounds__(T::NT)
gemm_x3_kernel(GemmOperand A, X3Weights B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg, int g_sk,
               int sk_base, int sk_rem, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN, AV = T::AV, BV = T::BV, ROWS = T::ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int fr = lane & 31, fh = lane >> 5;
  // fragment read offsets (bytes, inside one plane of one stage) of k16 group g: chunk = (2 g + fh) ^ ((row >> 2) & 3)
  const int fswz = (fr >> 2) & 3;
  int fa_off[2], fb_off[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int ch = ((2 * g + fh) ^ fswz) * 16;
    fa_off[g] = (wm * (BM / T::WM) + fr) * 64 + ch;
    fb_off[g] = (BM + wn * (BN / T::WN) + fr) * 64 + ch;
  }
  // staging roles.  A: float4 (4 k) of row (tid >> 3) + i * NT/8 -> 8 bytes of each plane.  B: 16-byte chunk.
  const int a_c = (tid & 7) >> 1, a_half = tid & 1;

  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);
  const int tiles_dp = dp_per_wg * G;
  const SkRange rg = blk < g_sk ? sk_range(blk, sk_base, sk_rem) : SkRange{0, 0};

  int dp_done = 0;
  for (int it = rg.begin; dp_done < dp_per_wg || it < rg.end;) {
    int tile, ks0, ks1;
    const bool dp = dp_done < dp_per_wg;
    if (dp) { tile = dp_done * G + blk; ks0 = 0; ks1 = ksteps; ++dp_done; }
    else {
      const int t = it / ksteps;
      tile = tiles_dp + t;
      ks0 = it - t * ksteps;
      ks1 = min(ksteps, ks0 + (rg.end - it));
    }
    const int nsteps = ks1 - ks0;
    int tile_m, tile_n;
    tile_origin<T::GROUP_N>(tile, tiles_m, tiles / tiles_m, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const float* pa[AV];
    int wa[AV];                                     // LDS byte offset of the 8-byte piece inside a plane
    int cy[AV], cx[AV];                             // A_CONV2: input row / column of tap (0, 0) for this output position
    (void)cy; (void)cx;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      if constexpr (T::AKIND == A_CONV2) {
        const int row = (tid >> 3) + i * (NT >> 3);
        const int g = m0 + row < M ? m0 + row : 0;
        const int pr = g / kUHW, pos = g - pr * kUHW, oy = pos / 7, ox = pos - oy * 7;
        cy[i] = oy - 1; cx[i] = ox - 1;
        pa[i] = A.ptr + (int64_t)pr * (128 * kUHW) + (tid & 7) * 4;
        wa[i] = row * 64 + ((a_c ^ ((row >> 2) & 3)) * 16) + a_half * 8;
      } else if constexpr (T::AKIND == A_UNION_FLAT) {
        // thread = (row tid % BM, k-group (tid / BM) * AV + i); rows past M read pair 0 (never stored by the epilogue)
        const int row = tid % BM, kg = (tid / BM) * AV + i;
        const int n = m0 + row < M ? m0 + row : 0;
        const int p = n / kUHW, hw = n - p * kUHW;
        pa[i] = A.ptr + (A.rowoff ? A.rowoff[p] : (int64_t)p * A.ld) + (int64_t)(ks0 * kBK + kg * 4) * kUHW + hw;
        wa[i] = row * 64 + (((kg >> 1) ^ ((row >> 2) & 3)) * 16) + (kg & 1) * 8;
      } else {
        const int row = (tid >> 3) + i * (NT >> 3);
        const int g = m0 + row;
        const int64_t* ro = A.rowoff;
        if (ro && A.aux > 0 && n0 >= A.aux) ro += M;         // grouped launch (GemmOperand::aux): second gather table
        pa[i] = A.ptr + (ro ? (g < M ? ro[g] : (int64_t)0)
                            : (int64_t)(g < M ? (A.rowidx ? A.rowidx[g] : g) : 0) * A.ld) + ks0 * kBK + (tid & 7) * 4;
        wa[i] = row * 64 + ((a_c ^ ((row >> 2) & 3)) * 16) + a_half * 8;
      }
    }
    const __bf16* pb[BV];
    int wb[BV];                                     // LDS byte offset inside the stage (plane included)
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int idx = tid + j * NT;
      const int plane = idx / (BN * 4), rem = idx - plane * (BN * 4), row = rem >> 2, c = rem & 3;
      const int g = n0 + row;
      pb[j] = B.planes + plane * B.plane_stride + (int64_t)(g < N ? g : 0) * B.ldp + ks0 * kBK + c * 8;
      wb[j] = (plane * ROWS + BM + row) * 64 + ((c ^ ((row >> 2) & 3)) * 16);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if constexpr (EpiInit<Epi>::value) {
      // C += A B (EpiUnionRows): the K range that starts a tile accumulates onto the output's old values
      if (ks0 == 0 && epi.init_on()) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = m0 + wm * (BM / T::WM) + fr + i * 32;
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int col = n0 + wn * (BN / T::WN) + 4 * fh + j * 32 + 8 * (e >> 2) + (e & 3);
              if (r < M && col < N) acc[i][j][e] = epi.init(r, col);
            }
        }
      }
    }

    // Two K-steps of global prefetch: the raw loads of step t+2 are issued at the top of step t into one of two register
    // sets; the set loaded one step earlier (step t+1) is split into planes and written to the other LDS stage under
    // the MFMAs of step t.
    f32x4 ra[2][AV];
    bf16x8 rb[2][BV];
    bool aok[2][AV];                                // A_CONV2: the piece's tap lies inside the image
    (void)aok;
    auto load_step = [&](int set, int step) {
      if (X3_ABLATE == 2) return;
      const int st = step < nsteps ? step : 0;            // steps past the range re-read step 0 (never consumed)
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        if constexpr (T::AKIND == A_CONV2) {
          const int ks = ks0 + st, tap = ks >> 2, ky = tap / 3, kx = tap - ky * 3;      // wave-uniform
          const int iy = cy[i] + ky, ix = cx[i] + kx;
          const bool ok = (unsigned)iy < 7u && (unsigned)ix < 7u;
          ra[set][i] = *reinterpret_cast<const f32x4*>(pa[i] + (ok ? (iy * 7 + ix) * 128 : 0) + (ks & 3) * kBK);
          aok[set][i] = ok;
        } else if constexpr (T::AKIND == A_UNION_FLAT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[set][i][e] = pa[i][(st * kBK + e) * kUHW];
        } else {
          ra[set][i] = *reinterpret_cast<const f32x4*>(pa[i] + st * kBK);
        }
      }
#pragma unroll
      for (int j = 0; j < BV; ++j) rb[set][j] = *reinterpret_cast<const bf16x8*>(pb[j] + st * kBK);
    };
    auto store_a = [&](int set, int i, unsigned char* stage) {
      bf16x4 h, m, l;
      if (X3_ABLATE == 2) return;
      if (X3_ABLATE == 1) {
        h = *reinterpret_cast<const bf16x4*>(&ra[set][i]); m = h; l = *(reinterpret_cast<const bf16x4*>(&ra[set][i]) + 1);
      } else if constexpr (T::AKIND == A_CONV2) {
        split3(aok[set][i] ? ra[set][i] : f32x4{0.f, 0.f, 0.f, 0.f}, h, m, l);
      } else
      split3(ra[set][i], h, m, l);
      *reinterpret_cast<bf16x4*>(stage + wa[i]) = h;
      *reinterpret_cast<bf16x4*>(stage + ROWS * 64 + wa[i]) = m;
      *reinterpret_cast<bf16x4*>(stage + 2 * ROWS * 64 + wa[i]) = l;
    };
    auto store_b = [&](int set, int j, unsigned char* stage) {
      if (X3_ABLATE == 2) return;
      *reinterpret_cast<bf16x8*>(stage + wb[j]) = rb[set][j];
    };
    auto read_frags = [&](const unsigned char* stage, int g, bf16x8 (&fa)[3][TM], bf16x8 (&fb)[3][TN]) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(stage + p * ROWS * 64 + fa_off[g] + i * 32 * 64);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(stage + p * ROWS * 64 + fb_off[g] + j * 32 * 64);
      }
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (a plane, b plane) of the six terms, small first
    // 6 x TM x TN MFMAs of one k16 group; `extra(n)` is called after the n-th MFMA (staging work rides in the gaps)
    auto mma_group = [&](const bf16x8 (&fa)[3][TM], const bf16x8 (&fb)[3][TN], auto&& extra) {
      int n = 0;
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {     // swapped ports: the weight fragment feeds the "A" port (EpiTraits)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB[t]][j], fa[PA[t]][i], acc[i][j], 0, 0, 0);
            extra(n++);
          }
    };

    load_step(0, 0);
#pragma unroll
    for (int i = 0; i < AV; ++i) store_a(0, i, smem_raw);
#pragma unroll
    for (int j = 0; j < BV; ++j) store_b(0, j, smem_raw);
    load_step(1, 1);
    __syncthreads();
    bf16x8 fa[3][TM], fb[3][TN];
    auto one_step = [&](int t, int set_store, int set_load) {
      const unsigned char* cur = smem_raw + (t & 1) * T::STAGE_BYTES;
      unsigned char* nxt = smem_raw + ((t + 1) & 1) * T::STAGE_BYTES;
      load_step(set_load, t + 2);            // the set written to LDS during the previous step receives step t+2
      read_frags(cur, 0, fa, fb);
      __builtin_amdgcn_sched_barrier(0);     // all twelve reads first: hipcc otherwise sinks each next to its MFMAs (12 exposed LDS latencies per step)
      // group 0: the split + LDS writes of step t+1 (raw data loaded during step t-1) ride under the MFMAs
      constexpr int NMG = 6 * TM * TN;
      mma_group(fa, fb, [&](int n) {                       // A pieces (split: VALU-heavy) spread over group 0
        if (n % (NMG / AV) == 0 && n / (NMG / AV) < AV) store_a(set_store, n / (NMG / AV), nxt);
      });
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 1, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      mma_group(fa, fb, [&](int n) {                       // B pieces over group 1
        if (n % (NMG / BV) == 0 && n / (NMG / BV) < BV) store_b(set_store, n / (NMG / BV), nxt);
      });
      __syncthreads();
    };
    {
      int t = 0;
      for (; t + 1 < nsteps; t += 2) {
        one_step(t, 1, 0);
        one_step(t + 1, 0, 1);
      }
      if (t < nsteps) one_step(t, 1, 0);
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
    __syncthreads();
  }
}

}  // namespace sttran
