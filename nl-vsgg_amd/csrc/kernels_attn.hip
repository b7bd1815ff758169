// kernels_attn.hip -- attention core, LayerNorm and row gather of the spatial encoder / temporal
// decoder (lib/transformer.py:20-30, 49-58; nn.MultiheadAttention semantics).
#include <algorithm>

#include "kernels.h"
#include "gemm_bf16x3_t16.h"

namespace sttran {

// ------------------------------------------------------------------------------------------
// Fused multi-head attention over ragged sequences.
//   qkv  [tokens][3*dim]  rows = [q | k | v] projections (bias already added)
//   out  [tokens][dim]    concat_h softmax(q_h k_h^T / sqrt(hd)) v_h
// One workgroup = (32-query tile, head, sequence), 4 wavefronts.
//   phase 1  S = Q K^T     per 32-key tile: K tile staged in LDS (head_dim zero-padded 242->256
//            so the MFMA K-loop needs no tail), four 16x16 quadrants, one per wave, on
//            v_mfma_f32_16x16x4_f32 with two interleaved accumulators
//   softmax  over the LDS score rows, max / sum reduced with wavefront shuffles
//   phase 2  O = P V       per 32-key tile: V tile staged in the same LDS buffer, P rows read as
//            the A operand (K-contiguous, ds_read_b128), V read as the B operand with
//            conflict-free ds_read_b32, 2 x (32x32) output tiles per wave on 32x32x2_f32
// Sequences are short here (<= 2 x boxes per frame), so K/V are re-staged per query tile.
// ------------------------------------------------------------------------------------------
constexpr int kHdPad = 256;           // padded head dim
constexpr int kQStride = kHdPad + 4;  // 260 dwords: rows land on distinct 4-dword LDS slots

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void stage_head_rows(float* dst, const float* __restrict__ src, int64_t ld,
                                                int row0, int nrows_valid, int hd, float scale, int tid) {
  // 32 rows x kQStride; 8 lanes per row, float2 granules (rows are only 8-byte aligned: hd*4 = 968)
  const int r = tid >> 3, t = tid & 7;
  const bool rv = r < nrows_valid;
  const float* s = src + (int64_t)(row0 + r) * ld;
  float* d = dst + r * kQStride;
  for (int i = t; i < kQStride / 2; i += 8) {
    f32x2 v = {0.f, 0.f};
    if (rv && 2 * i < hd) v = *reinterpret_cast<const f32x2*>(s + 2 * i);
    v[0] *= scale; v[1] *= scale;
    *reinterpret_cast<f32x2*>(d + 2 * i) = v;
  }
}

// Keys are processed in chunks of `kc` (<= kAttnChunkKeys, what the LDS score block holds) with a running row maximum
// and row sum (online softmax): a sequence of ANY length is handled -- the reference has no limit
// (lib/transformer.py:130-163 pads to the longest frame / window whatever it is).  A sequence that fits one chunk
// (<= 480 keys: everything this kernel saw in rounds 1-2) takes exactly the former single-pass arithmetic.
__global__ void __launch_bounds__(256)
attention_kernel(const float* __restrict__ qkv, const int* __restrict__ seq_off, const int* __restrict__ seq_len,
                 float* __restrict__ out, int64_t ldo, int dim, int hd, float scale, int kc, int len_lo, int len_hi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int s = blockIdx.z, h = blockIdx.y;
  const int L = seq_len[s];
  // (len_lo, len_hi]: shorter sequences belong to another launch (launch_attention_classes)
  if (L <= len_lo || L > len_hi) return;
  // query tiles of this workgroup: blockIdx.x, + gridDim.x, ...  launch_attention gives every 32-query tile its own
  // workgroup; launch_attention_classes (lengths known on the device only: most slots of a launch are empty or belong to
  // another class) launches ONE workgroup per (head, slot) that walks the tiles -- an empty slot then costs one workgroup,
  // not ceil(bound / 32) of them (DSG-DETR: the all-empty (80, 176] launch 54 -> 9 us, three of them per step)
  for (int q0 = blockIdx.x * 32; q0 < L; q0 += gridDim.x * 32) {
  const int base = seq_off[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ps = kc + 4;                        // score row stride
  float* Qs = smem;                             // [32][260]
  float* KVs = Qs + 32 * kQStride;              // [32][260]
  float* Ps = KVs + 32 * kQStride;              // [32][kc+4]
  float* rmax = Ps + 32 * ps;                   // [32] running maximum of each query row
  float* rsum = rmax + 32;                      // [32] running sum of exp(score - rmax)
  float* ralpha = rsum + 32;                    // [32] exp(old maximum - new maximum) of the current chunk
  const int64_t ld = 3 * (int64_t)dim;
  const float* qp = qkv + (int64_t)base * ld + h * hd;

  stage_head_rows(Qs, qp, ld, q0, min(32, L - q0), hd, scale, tid);
  if (tid < 32) { rmax[tid] = -INFINITY; rsum[tid] = 0.f; }

  const int qi = wave >> 1, kj = wave & 1, l15 = lane & 15, g = lane >> 4;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  const int d0 = wave * 64;                     // this wave's two 32-column output tiles

  for (int k0 = 0; k0 < L; k0 += kc) {
    const int Lc = min(kc, L - k0);             // keys of this chunk
    const int nkt = (Lc + 31) / 32;
    // ---- phase 1: scores of the chunk -----------------------------------------------------
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();                          // previous tile's reads of KVs are done
      stage_head_rows(KVs, qp + dim, ld, k0 + kt * 32, min(32, Lc - kt * 32), hd, 1.f, tid);
      __syncthreads();
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
      const float* ar = Qs + (qi * 16 + l15) * kQStride + 4 * g;
      const float* br = KVs + (kj * 16 + l15) * kQStride + 4 * g;
#pragma unroll 4
      for (int kb = 0; kb < kHdPad / 16; ++kb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ar + kb * 16);
        const f32x4 b = *reinterpret_cast<const f32x4*>(br + kb * 16);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c1, 0, 0, 0);
      }
      // C/D of 16x16: col = lane&15 (key), row = 4*(lane>>4) + e (query)
#pragma unroll
      for (int e = 0; e < 4; ++e) Ps[(qi * 16 + 4 * g + e) * ps + kt * 32 + kj * 16 + l15] = c0[e] + c1[e];
    }
    __syncthreads();

    // ---- softmax over the chunk's keys of each of the 32 rows; 8 rows per wave ------------------
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {
      float* pr = Ps + r * ps;
      float m = rmax[r];
      const float m_old = m;
      for (int c = lane; c < Lc; c += 64) m = fmaxf(m, pr[c]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
      for (int c = lane; c < nkt * 32; c += 64) {
        const float e = (c < Lc) ? expf(pr[c] - m) : 0.f;
        pr[c] = e;
        sum += e;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0) {
        const float alpha = k0 ? expf(m_old - m) : 0.f;      // first chunk: nothing accumulated yet
        ralpha[r] = alpha;
        rmax[r] = m;
        rsum[r] = k0 ? rsum[r] * alpha + sum : sum;
      }
    }

    // ---- phase 2: O = O * alpha + P V over the chunk --------------------------------------------
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();                          // (first tile: also publishes the softmax results)
      stage_head_rows(KVs, qp + 2 * dim, ld, k0 + kt * 32, min(32, Lc - kt * 32), hd, 1.f, tid);
      if (kt == 0 && k0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float al = ralpha[(e & 3) + 8 * (e >> 2) + 4 * fh];
          o0[e] *= al; o1[e] *= al;
        }
      }
      __syncthreads();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Ps + fr * ps + kt * 32 + kb * 8 + 4 * fh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* vr = KVs + (kb * 8 + 4 * fh + e) * kQStride + d0 + fr;
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], vr[0], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], vr[32], o1, 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  float* op = out + (int64_t)(base + q0) * ldo + h * hd;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int q = (e & 3) + 8 * (e >> 2) + 4 * fh;
    if (q0 + q < L) {
      const float ri = 1.f / rsum[q];
      const int c0 = d0 + fr, c1 = d0 + 32 + fr;
      if (c0 < hd) op[(int64_t)q * ldo + c0] = o0[e] * ri;
      if (c1 < hd) op[(int64_t)q * ldo + c1] = o1[e] * ri;
    }
  }
  __syncthreads();                              // rsum / Qs are rewritten by the next query tile
  }
}

// ------------------------------------------------------------------------------------------
// Short-sequence attention (every sequence <= kAttnShortMax keys -- always the case for per-frame /
// two-frame-window sequences of a few dozen boxes): ONE workgroup per (sequence, head) stages K and V
// once for all of the sequence's queries, on 16x16x4 MFMA tiles (16-row granularity: a 70-token
// window pads to 80, not 96).
//   phase 1  S = Q K^T : head_dim is processed in four 64-wide chunks so that Q and K chunks
//            ([L16][68] each, row stride 68 dwords = conflict-free ds_read_b128) fit LDS together;
//            each wave owns up to 7 score tiles whose accumulators persist across the chunks
//   softmax  rows of S in LDS, wavefront shuffles, P written back normalised
//   phase 2  O = P V   : V overwrites the Q/K region, one 128-column half at a time ([L16][132]); wave w
//            owns columns [32w, 32w+32) of each half for all query tiles
// LDS: 70 KB at 80 keys -> two workgroups per CU, so one stages while the other runs MFMAs. (P rows as A operand via ds_read_b128, V as B operand via
//            conflict-free ds_read_b32)
// q_begin (optional) = first query row of each sequence to compute: the last decoder layer only
// needs the rows the 'latter' scatter reads (lib/transformer.py:179-185).
// ------------------------------------------------------------------------------------------
constexpr int kAttnShortMax = 80;

// Staging of a [rows][NCOLS_PAD] block in two halves so that the global loads of the NEXT block can be in flight
// while the MFMAs of the current one run: stage_load4_nb pulls this thread's 16-byte granules into registers (zero for
// rows / columns outside the matrix), stage_store4 scales and writes them to LDS.  256 threads; MAXU4 = most granules per
// thread (compile time: the arrays stay in VGPRs).
// Rows of a head are only 8-byte aligned (head_dim * 4 = 968 bytes): V4a8 = four floats at such an address -- global
// memory takes a 16-byte access there (unaligned access mode; hipcc emits global_load / store_dwordx4 for it).
// No branches: a granule outside the matrix reads the block's first element (always there) and is zeroed by selects --
// with guarded loads hipcc carried the register arrays through the branches as whole tuples and spilled them.
// A granule is all in, all out, or -- the last one of a head whose dimension is 4 k + 2 -- half in: that one is fetched two
// floats earlier (inside the head) and shifted, so nothing past the head is ever read.
struct __attribute__((packed, aligned(8))) V4a8 { f32x4 v; };

template <int NCOLS_PAD, int MAXU4>
__device__ __forceinline__ void stage_load4_nb(f32x4 (&v)[MAXU4], const float* __restrict__ src, int64_t ld, int nrows_valid,
                                               int col0, int ncols_valid, int tid) {
  constexpr int Q = NCOLS_PAD / 4;
#pragma unroll
  for (int u = 0; u < MAXU4; ++u) {
    const int i = tid + u * 256, r = i / Q, c = col0 + (i % Q) * 4;
    const bool row = r < nrows_valid, full = row && c + 3 < ncols_valid, half = row && !full && c + 1 < ncols_valid;
    const f32x4 x = reinterpret_cast<const V4a8*>((full || half) ? src + (int64_t)r * ld + (half ? c - 2 : c) : src)->v;
    v[u] = f32x4{full ? x[0] : (half ? x[2] : 0.f), full ? x[1] : (half ? x[3] : 0.f), full ? x[2] : 0.f, full ? x[3] : 0.f};
  }
}
template <int NCOLS_PAD, int MAXU4>
__device__ __forceinline__ void stage_store4(float* dst, int dstride, const f32x4 (&v)[MAXU4], int nrows_pad, float scale,
                                             int tid) {
  constexpr int Q = NCOLS_PAD / 4;
  const int total = nrows_pad * Q;
#pragma unroll
  for (int u = 0; u < MAXU4; ++u) {
    const int i = tid + u * 256, r = i / Q, c4 = (i % Q) * 4;
    if (i < total) *reinterpret_cast<f32x4*>(dst + r * dstride + c4) = v[u] * scale;
  }
}
// CHUNK = head-dim chunk of phase 1, VW = V columns staged per pass of phase 2: smaller chunks = more barriers and
// round trips per workgroup, but less LDS and fewer registers, i.e. more resident workgroups to cover them.  Which
// instantiation serves which launch: short_variant() below (round 5: residency wins wherever the launch fills the chip).
template <int CHUNK, int VW, int MAXROWS, int MINWG, bool EARLY = false>
__global__ void __launch_bounds__(256, MINWG)
attention_short_kernel(const float* __restrict__ qkv, const int* __restrict__ seq_off,
                       const int* __restrict__ seq_len, const int* __restrict__ q_begin, float* __restrict__ out,
                       int64_t ldo, int dim, int hd, float scale, int l16max, int len_lo, int len_hi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kMaxQ = MAXROWS / 16;                   // query (and key) tiles of 16 rows
  constexpr int TPW = (kMaxQ * kMaxQ + 3) / 4;          // score tiles per wave
  const int s = blockIdx.y, h = blockIdx.x;
  const int L = seq_len[s];
  if (L <= len_lo || L > len_hi) return;                // (len_lo, len_hi]: this launch's length class; len_lo >= 0
  const int qb = q_begin ? q_begin[s] : 0;
  const int Lq = L - qb;
  if (Lq <= 0) return;
  const int base = seq_off[s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Lk16 = (L + 15) & ~15, Lq16 = (Lq + 15) & ~15;
  const int nq = Lq16 >> 4, nk = Lk16 >> 4, ntiles = nq * nk;
  constexpr int kChunk = CHUNK, kCStride = CHUNK + 4;   // 68 / 132 dwords: (stride/4) odd -> distinct LDS slots
  constexpr int kVHalf = VW, kVStride = VW + 4;
  constexpr int NVH = kHdPad / VW, DTW = VW / 64;       // V passes; 16-column tiles per wave per pass
  const int region = max(2 * l16max * kCStride, l16max * kVStride);
  float* Qc = smem;                          // [Lq16][CHUNK+4]
  float* Kc = smem + l16max * kCStride;      // [Lk16][CHUNK+4]
  float* Vs = smem;                          // [Lk16][VW+4]   (phase 2, overwrites Qc/Kc)
  float* Ps = smem + region;                 // [Lq16][Lk16 + 4]
  const int ps = Lk16 + 4;
  const int64_t ld = 3 * (int64_t)dim;
  const float* qp = qkv + (int64_t)base * ld + h * hd;

  // ---- phase 1: S = (Q * scale) K^T over head-dim chunks.  The global loads of chunk c+1 (and, behind the last
  //      chunk, of the first V block) are issued before the MFMAs of chunk c and written to LDS after them, so
  //      only the very first load latency of a workgroup is exposed.
  constexpr int NCH = kHdPad / kChunk;
  constexpr int UQ4 = (MAXROWS * (kChunk / 4) + 255) / 256;     // 16-byte granules per thread of a Q or K chunk
  constexpr int UV4 = (MAXROWS * (kVHalf / 4) + 255) / 256;     // ... of a V block
  const float* vp = qp + 2 * dim;
  f32x4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 rv[UV4];
  auto score_tiles = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < ntiles) {
        const int qi = tile / nk, kj = tile - qi * nk;
        const float* ar = Qc + (qi * 16 + l15) * kCStride + 4 * g;
        const float* br = Kc + (kj * 16 + l15) * kCStride + 4 * g;
        f32x4 c0a = acc[t], c1a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < kChunk / 16; ++kb) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(ar + kb * 16);
          const f32x4 b = *reinterpret_cast<const f32x4*>(br + kb * 16);
          c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c0a, 0, 0, 0);
          c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c1a, 0, 0, 0);
          c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c0a, 0, 0, 0);
          c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c1a, 0, 0, 0);
        }
        acc[t] = c0a + c1a;
      }
    }
  };
  // C/D of 16x16: col = lane&15 (key), row = 4*(lane>>4) + e (query)
  auto write_scores = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < ntiles) {
        const int qi = tile / nk, kj = tile - qi * nk;
#pragma unroll
        for (int e = 0; e < 4; ++e) Ps[(qi * 16 + 4 * g + e) * ps + kj * 16 + l15] = acc[t][e];
      }
    }
  };
  if constexpr (EARLY) {
    // sequences of <= 32 keys: every operand of the workgroup is 96 registers per thread, so ALL global loads -- both Q / K
    // chunks and the V block -- are issued before anything else and the workgroup waits for memory once instead of once
    // per chunk (with 22 keys the MFMAs of a chunk are far too short to cover the next chunk's round trip)
    static_assert(NCH == 2, "two head-dim chunks");
    static_assert(NVH == 1, "one V block");
    f32x4 rq0[UQ4], rk0[UQ4], rq1[UQ4], rk1[UQ4];
    stage_load4_nb<kChunk, UQ4>(rq0, qp + (int64_t)qb * ld, ld, Lq, 0, hd, tid);
    stage_load4_nb<kChunk, UQ4>(rk0, qp + dim, ld, L, 0, hd, tid);
    stage_load4_nb<kChunk, UQ4>(rq1, qp + (int64_t)qb * ld, ld, Lq, kChunk, hd, tid);
    stage_load4_nb<kChunk, UQ4>(rk1, qp + dim, ld, L, kChunk, hd, tid);
    stage_load4_nb<kVHalf, UV4>(rv, vp, ld, L, 0, hd, tid);
    stage_store4<kChunk, UQ4>(Qc, kCStride, rq0, Lq16, scale, tid);
    stage_store4<kChunk, UQ4>(Kc, kCStride, rk0, Lk16, 1.f, tid);
    __syncthreads();
    score_tiles();
    __syncthreads();                                // every wave is done reading the first chunk
    stage_store4<kChunk, UQ4>(Qc, kCStride, rq1, Lq16, scale, tid);
    stage_store4<kChunk, UQ4>(Kc, kCStride, rk1, Lk16, 1.f, tid);
    __syncthreads();
    score_tiles();
    // scores -> LDS, then V over the Q / K region (every wave has read its last fragments: barrier)
    write_scores();
    __syncthreads();
    stage_store4<kVHalf, UV4>(Vs, kVStride, rv, Lk16, 1.f, tid);
  } else {
  f32x4 rq[UQ4], rk[UQ4];
  stage_load4_nb<kChunk, UQ4>(rq, qp + (int64_t)qb * ld, ld, Lq, 0, hd, tid);
  stage_load4_nb<kChunk, UQ4>(rk, qp + dim, ld, L, 0, hd, tid);
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    if (ci) __syncthreads();                      // every wave is done reading the previous chunk
    stage_store4<kChunk, UQ4>(Qc, kCStride, rq, Lq16, scale, tid);
    stage_store4<kChunk, UQ4>(Kc, kCStride, rk, Lk16, 1.f, tid);
    __syncthreads();
    if (ci + 1 < NCH) {
      stage_load4_nb<kChunk, UQ4>(rq, qp + (int64_t)qb * ld, ld, Lq, (ci + 1) * kChunk, hd, tid);
      stage_load4_nb<kChunk, UQ4>(rk, qp + dim, ld, L, (ci + 1) * kChunk, hd, tid);
    } else {
      stage_load4_nb<kVHalf, UV4>(rv, vp, ld, L, 0, hd, tid);
    }
    __builtin_amdgcn_sched_barrier(0);            // keep the loads ahead of the MFMAs
    score_tiles();
  }
  write_scores();
  __syncthreads();
  // the first V block (already in registers) goes to LDS first; the second lands during softmax + PV
  stage_store4<kVHalf, UV4>(Vs, kVStride, rv, Lk16, 1.f, tid);
  if (NVH > 1) stage_load4_nb<kVHalf, UV4>(rv, vp, ld, L, kVHalf, hd, tid);
  }

  // ---- softmax rows (wavefront shuffles) ------------------------------------------------------------------------------
  for (int r = wave; r < Lq16; r += 4) {
    float* pr = Ps + r * ps;
    float m = -INFINITY;
    for (int c = lane; c < L; c += 64) m = fmaxf(m, pr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e0 = (lane < L) ? expf(pr[lane] - m) : 0.f;
    float e1 = (lane + 64 < L) ? expf(pr[lane + 64] - m) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float ri = 1.f / sum;
    if (lane < Lk16) pr[lane] = e0 * ri;
    if (lane + 64 < Lk16) pr[lane + 64] = e1 * ri;
  }
  __syncthreads();

  // ---- phase 2: O[q][d] = sum_key P[q][key] V[key][d]; per V half, wave owns d in [32w, 32w+32) -------
  f32x4 o[kMaxQ][4];                             // [query tile][pass*DTW + d-tile]
#pragma unroll
  for (int qi = 0; qi < kMaxQ; ++qi)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int vh = 0; vh < NVH; ++vh) {
    if (vh) {
      __syncthreads();                           // every wave is done with the previous block
      stage_store4<kVHalf, UV4>(Vs, kVStride, rv, Lk16, 1.f, tid);
      if (vh + 1 < NVH) stage_load4_nb<kVHalf, UV4>(rv, vp, ld, L, (vh + 1) * kVHalf, hd, tid);
      __syncthreads();
    }
    for (int kb = 0; kb < nk; ++kb) {
      float bv[DTW][4];
#pragma unroll
      for (int dt = 0; dt < DTW; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[dt][e] = Vs[(kb * 16 + 4 * g + e) * kVStride + wave * (16 * DTW) + dt * 16 + l15];
#pragma unroll
      for (int qi = 0; qi < kMaxQ; ++qi) {
        if (qi < nq) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(Ps + (qi * 16 + l15) * ps + kb * 16 + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int dt = 0; dt < DTW; ++dt)
              // V on the "A" port, P on the "B" port: the block comes out transposed, O^T[d][q] -- a lane then holds FOUR
              // CONSECUTIVE d of one query row (two 8-byte stores) instead of one d of four query rows (four dword stores)
              o[qi][vh * DTW + dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[dt][e], a[e], o[qi][vh * DTW + dt], 0, 0, 0);
        }
      }
    }
  }
  // C/D of the transposed block: col = lane & 15 = query row, row = 4 (lane >> 4) + e = d.  Rows of `out` are 16-byte
  // aligned, a head starts at a multiple of 8 bytes (head_dim is even)
  float* op = out + (int64_t)(base + qb) * ldo + h * hd;
#pragma unroll
  for (int qi = 0; qi < kMaxQ; ++qi) {
    const int q = qi * 16 + l15;
    if (qi < nq && q < Lq) {
      float* orow = op + (int64_t)q * ldo;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = (dt / DTW) * kVHalf + wave * (16 * DTW) + (dt % DTW) * 16 + 4 * g;
        // one 16-byte store at an 8-byte aligned address (global memory takes it: unaligned access mode); the last
        // group of a head (head_dim 242 = 60 x 4 + 2) is half in
        if (d + 3 < hd) reinterpret_cast<V4a8*>(orow + d)->v = o[qi][dt];
        else if (d < hd) *reinterpret_cast<f32x2*>(orow + d) = f32x2{o[qi][dt][0], o[qi][dt][1]};
      }
    }
  }
}

// dynamic-LDS limits already raised per device, shared by both launchers (a second set of marks could LOWER a limit)
static DeviceMarks g_attn_marks[5], g_attn_marks_long;

// The short-sequence variants (round 5: chosen by LENGTH and by how many workgroups the launch has).
//   V_E32  <128,256,32,4,EARLY>  <= 32 keys, every global load issued up front: ONE memory wait per workgroup -- the lowest
//          latency, 38 KB of LDS and 123 registers = four workgroups per CU.  Kept where latency decides: launches with too few
//          workgroups to fill the chip (one clip per call), and full 25 .. 32-key sequences (660 x 32: 142 vs 149 us).
//   V_C32  <64,128,32,8>         <= 24 keys in a launch that fills the chip: head-dim chunks of 64, V in two halves -- 22 KB of
//          LDS and 60 registers = SEVEN workgroups per CU; more round trips per workgroup, but with ~22 keys a workgroup is
//          nothing but round trips and the extra residents cover them: 1 024 x 11 keys 99 -> 84 us, 960 x 22 162 -> 153 us in
//          situ (ms_attention 0.541 -> 0.504 per step).
//   V_C48  <64,128,48,4>         33 .. 48 keys, always: 36 KB of LDS instead of the 60 KB of the <128,256> form it replaces, i.e.
//          four workgroups per CU instead of two: 256 x 35 keys (the 64x36 encoder) 120 -> 90 us.
//   V_80   <32,64,80,3>          49 .. 80 keys: head-dim chunks of 32, V in quarters -- 50 KB of LDS (the 27 KB score block
//          included) instead of the 70 KB of <64,128>, three workgroups per CU instead of two: 252 x 70 keys (the 64x36
//          decoder windows) 243 -> 230 us.
enum { V_E32 = 0, V_C32 = 1, V_C48 = 2, V_80 = 3 };
static int short_variant(int len, int64_t workgroups) {
  if (len > 48) return V_80;
  if (len > 32) return V_C48;
  return (len <= 24 && workgroups >= (int64_t)7 * std::max(num_cus(), 1)) ? V_C32 : V_E32;
}
static int short_lds_bytes(int variant, int l16) {
  const int cs = (variant == V_E32 ? 128 : variant == V_80 ? 32 : 64) + 4, vs = (variant == V_E32 ? 256 : variant == V_80 ? 64 : 128) + 4;
  return (std::max(2 * l16 * cs, l16 * vs) + l16 * (l16 + 4)) * 4;
}
using ShortKernel = void (*)(const float*, const int*, const int*, const int*, float*, int64_t, int, int, float, int, int, int);
static ShortKernel short_kernel(int variant) {
  switch (variant) {
    case V_E32: return attention_short_kernel<128, 256, 32, 4, true>;
    case V_C32: return attention_short_kernel<64, 128, 32, 8>;
    case V_C48: return attention_short_kernel<64, 128, 48, 4>;
    default: return attention_short_kernel<32, 64, kAttnShortMax, 3>;
  }
}

hipError_t launch_attention(hipStream_t s, const float* qkv, const int* seq_off, const int* seq_len,
                            const int* q_begin, int num_seq, int max_len, float* out, int64_t ldo, int dim, int nhead) {
  if (num_seq <= 0 || max_len <= 0) return hipSuccess;
  const int hd = dim / nhead;
  if (hd > kHdPad - 2 || (hd & 1) || hd < 4 || (ldo & 1)) return hipErrorInvalidValue;      // 8-byte accesses of rows and heads
  const float scale = 1.0f / sqrtf((float)hd);
  if (max_len <= kAttnShortMax) {
    const int l16 = (max_len + 15) & ~15;
    const int var = short_variant(max_len, (int64_t)num_seq * nhead);
    const int lds = short_lds_bytes(var, l16);
    auto kern = short_kernel(var);
    hipError_t e = g_attn_marks[var].raise_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nhead, num_seq), dim3(256), lds, s, qkv, seq_off, seq_len, q_begin, out, ldo, dim,
                       hd, scale, l16, 0, 1 << 30);
    return hipGetLastError();
  }
  // the long-sequence kernel computes every query row: q_begin is an optimisation hint only (rows before
  // it are never read by the caller), so it is simply not used here
  const int kc = std::min((max_len + 31) / 32 * 32, kAttnChunkKeys);      // keys per pass of the LDS score block
  const int lds = (2 * 32 * kQStride + 32 * (kc + 4) + 96) * 4;
  hipError_t e = g_attn_marks_long.raise_lds(reinterpret_cast<const void*>(attention_kernel), lds);
  if (e != hipSuccess) return e;
  dim3 grid((max_len + 31) / 32, nhead, num_seq);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, s, qkv, seq_off, seq_len, out, ldo, dim, hd, scale, kc, 0, 1 << 30);
  return hipGetLastError();
}

// The same attention when the host does NOT know the sequence lengths (DSG-DETR's class sequences are built on the
// device): `len_bound` >= every length.  One launch per length class the bound allows -- (0, 32] and (32, 80] on the
// short-sequence variants, (80, bound] on the general kernel with one workgroup per (head, slot) --, each over all `num_seq`
// slots; a workgroup whose sequence is empty or belongs to another class returns at once.  No read-back, capturable.
hipError_t launch_attention_classes(hipStream_t s, const float* qkv, const int* seq_off, const int* seq_len, int num_seq,
                                    int len_bound, float* out, int64_t ldo, int dim, int nhead) {
  if (num_seq <= 0 || len_bound <= 0) return hipSuccess;
  const int hd = dim / nhead;
  if (hd > kHdPad - 2 || (hd & 1) || hd < 4 || (ldo & 1)) return hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)hd);
  // classes: (0, 32], (32, 80] on the 80-key variant (one launch for both upper short classes: in the launches this entry
  // point serves they are empty or nearly so, and an empty launch costs ~9 us), (80, bound] on the general kernel
  const int edges[3] = {0, 32, kAttnShortMax};
  for (int v = 0; v < 2; ++v) {
    if (len_bound <= edges[v]) break;
    const int hi = edges[v + 1];
    const int top = std::min(len_bound, hi), l16 = (top + 15) & ~15;
    // The <= 32 class: the bound says little here (DSG-DETR's bound is the pairs of a clip, its class sequences average
    // pairs / classes ~ 5 tokens): a launch with enough slots to fill the chip takes the high-residency form whatever the
    // bound (correct up to 32 keys like V_E32; full 25..32-key sequences would be ~5 % slower on it), a small one V_E32
    const bool dense = (int64_t)num_seq * nhead >= (int64_t)7 * std::max(num_cus(), 1);
    // (one rule with launch_attention: `short_variant` picks V_C32 for <= 24 keys only when the launch fills the chip --
    // a small launch is latency-bound and takes V_E32 there as well; `dense` extends V_C32 to bounds of 25..32)
    const int var = v == 0 ? (dense ? V_C32 : V_E32) : (top <= 48 ? V_C48 : V_80);
    const int lds = short_lds_bytes(var, l16);
    auto kern = short_kernel(var);
    hipError_t e = g_attn_marks[var].raise_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nhead, num_seq), dim3(256), lds, s, qkv, seq_off, seq_len, (const int*)nullptr, out, ldo, dim,
                       hd, scale, l16, edges[v], hi);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (len_bound > kAttnShortMax) {
    const int kc = std::min((len_bound + 31) / 32 * 32, kAttnChunkKeys);
    const int lds = (2 * 32 * kQStride + 32 * (kc + 4) + 96) * 4;
    hipError_t e = g_attn_marks_long.raise_lds(reinterpret_cast<const void*>(attention_kernel), lds);
    if (e != hipSuccess) return e;
    // workgroups per (head, slot): each walks its query tiles with stride gridDim.x.  ONE when the launch already fills the
    // chip (DSG-DETR at 16x12: hundreds of slots, all empty in this class -- an empty workgroup costs nothing, and one per
    // slot keeps the launch short); a few when there are only a handful of slots with really long sequences, so that their
    // ceil(L / 32) tiles do not run one after the other on nhead x few workgroups
    const int tiles = (len_bound + 31) / 32;
    const int64_t slots = (int64_t)nhead * num_seq;
    const int gx = (int)std::min<int64_t>(tiles, std::max<int64_t>(1, 2 * (int64_t)std::max(num_cus(), 1) / slots));
    dim3 grid(gx, nhead, num_seq);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, s, qkv, seq_off, seq_len, out, ldo, dim, hd, scale, kc,
                       kAttnShortMax, len_bound);
    return hipGetLastError();
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim (biased variance, eps 1e-5): one wavefront per row, the row held in
// registers between the mean, variance and normalise passes; reductions by wavefront shuffles.
// ------------------------------------------------------------------------------------------
constexpr int kLnMaxV = 16;   // float4 per lane -> dim <= 4096

// PLANES (bf16x3 engine): the normalised row additionally leaves as fragment-major bf16 planes (gemm_bf16x3_t16.h) -- the
// activation operand of the GEMM behind the LayerNorm (linear1, the next layer's in_proj) without a split pass of its own
template <bool PLANES>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                 const float* __restrict__ beta, float* __restrict__ y, int64_t ldy, int64_t rows, int dim,
                 __bf16* __restrict__ planes, int kb_total) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = dim >> 2;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ldx);
  f32x4 v[kLnMaxV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxV; ++i) {
    const int j = lane + 64 * i;
    v[i] = (j < n4) ? xr[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxV; ++i) {
    const int j = lane + 64 * i;
    if (j < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)dim + 1e-5f);
  f32x4* yr = reinterpret_cast<f32x4*>(y + row * ldy);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
  for (int i = 0; i < kLnMaxV; ++i) {
    const int j = lane + 64 * i;
    if (j < n4) {
      const f32x4 g = g4[j], b = b4[j];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      yr[j] = o;
      if constexpr (PLANES) fm_store4(planes, kb_total, (int)row, 4 * j, o);
    } else if constexpr (PLANES) {
      if (4 * j < kb_total * 32) fm_store4(planes, kb_total, (int)row, 4 * j, f32x4{0.f, 0.f, 0.f, 0.f});   // the K tail stays zero
    }
  }
}

hipError_t launch_layernorm(hipStream_t s, const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                            int64_t ldy, int64_t rows, int dim, void* planes) {
  if (rows <= 0) return hipSuccess;
  if ((dim & 3) || dim > kLnMaxV * 256 || (ldx & 3) || (ldy & 3) || ldx < dim || ldy < dim) return hipErrorInvalidValue;
  const int kb = (dim + 31) / 32;
  if (planes) {
    if (kb * 32 > kLnMaxV * 256 || rows >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy,
                       rows, dim, reinterpret_cast<__bf16*>(planes), kb);
  } else {
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy,
                       rows, dim, (__bf16*)nullptr, kb);
  }
  return hipGetLastError();
}

// dst[r, :] = src[idx[r], :]   (window build lib/transformer.py:153, 'latter' scatter :181-185)
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, int64_t lds4, const int* __restrict__ idx, float* __restrict__ dst,
                   int64_t ldd4, int64_t rows, int n4) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const f32x4* s = reinterpret_cast<const f32x4*>(src) + (int64_t)idx[row] * lds4;
  f32x4* d = reinterpret_cast<f32x4*>(dst) + row * ldd4;
  for (int j = threadIdx.x; j < n4; j += 256) d[j] = s[j];
}

hipError_t launch_gather_rows(hipStream_t s, const float* src, int64_t lds, const int* idx, float* dst, int64_t ldd,
                              int64_t rows, int dim) {
  if (rows <= 0) return hipSuccess;
  if ((dim & 3) || (lds & 3) || (ldd & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, lds >> 2, idx, dst, ldd >> 2, rows,
                     dim >> 2);
  return hipGetLastError();
}

// dst[r, :] = src[idx[r], :] + table[tidx[r], :]   (DSG-DETR: sequence gather + sinusoidal PE,
// lib/dsg_detr.py:42-46,556-559)
__global__ void __launch_bounds__(256)
gather_add_rows_kernel(const float* __restrict__ src, int64_t lds4, const int* __restrict__ idx,
                       const float* __restrict__ table, int64_t ldt4, const int* __restrict__ tidx,
                       float* __restrict__ dst, int64_t ldd4, int64_t rows, int n4) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const f32x4* s = reinterpret_cast<const f32x4*>(src) + (int64_t)idx[row] * lds4;
  const f32x4* t = reinterpret_cast<const f32x4*>(table) + (int64_t)tidx[row] * ldt4;
  f32x4* d = reinterpret_cast<f32x4*>(dst) + row * ldd4;
  for (int j = threadIdx.x; j < n4; j += 256) d[j] = s[j] + t[j];
}

hipError_t launch_gather_add_rows(hipStream_t s, const float* src, int64_t lds, const int* idx, const float* table,
                                  int64_t ldt, const int* tidx, float* dst, int64_t ldd, int64_t rows, int dim) {
  if (rows <= 0) return hipSuccess;
  if ((dim & 3) || (lds & 3) || (ldt & 3) || (ldd & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_add_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, lds >> 2, idx, table, ldt >> 2,
                     tidx, dst, ldd >> 2, rows, dim >> 2);
  return hipGetLastError();
}

}  // namespace sttran
