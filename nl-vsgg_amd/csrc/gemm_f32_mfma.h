// gemm_f32_mfma.h -- exact-fp32 MFMA GEMM for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        A rows optionally gathered through rowidx
//
// Every dense contraction of the STTran path (nn.Linear: lib/sttran.py:346-348,370-372,
// lib/transformer.py:9-12,38-42; conv3x3 as implicit GEMM: lib/sttran.py:342) runs through this
// kernel.  Parity with the fp32 reference is 1e-3, so the matrix pipe is driven with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate -- bit-identical to an fmaf chain), whose peak
// is 64 FLOP/clk/SIMD = 157.3 TFLOP/s per MI355X.
//
// Structure (one workgroup = WM x WN wavefronts of 64 lanes, block tile BM x BN, BK = 32):
//   * both operands are K-contiguous rows ("NT" GEMM), staged global -> VGPR -> LDS as whole
//     128-byte row segments (8 lanes x dwordx4 per row: full cache lines);
//   * LDS rows are padded to 36 dwords, which makes the ds_read_b128 fragment reads
//     conflict-free (36*i mod 64 hits 16 distinct 4-dword slots for any 16 rows);
//   * one ds_read_b128 per lane feeds FOUR MFMAs: lane (r, h) holds k = kb+4h+{0..3}; MFMA j
//     consumes element j of both fragments, i.e. k = kb+j and kb+4+j -- the k order inside a
//     group of 8 is permuted identically for A and B, which leaves the dot product unchanged;
//   * double-buffered LDS, global loads of tile t+1 are issued before the MFMAs of tile t and
//     written to the other buffer after them: one barrier per K-step;
//   * persistent workgroups on a stream-K schedule (equal MFMA work per CU, see below), remapped
//     so that workgroups sharing a weight panel sit on one XCD (its L2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sttran {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// torch.relu and torch's max_pool2d propagate NaN; fmaxf / v_max_f32 return the other operand.  A clip with a
// non-finite input must come out non-finite (as it does from the reference), not silently cleaned.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a < b || b != b) ? b : a; }

constexpr int kBK = 32;        // K-step
constexpr int kLdsStride = 36; // dwords per staged row (32 + 4 pad)

struct GemmOperand {
  const float* ptr;
  int64_t ld;          // row stride in floats (multiple of 4, 16-byte aligned base)
  const int* rowidx;   // optional gather: logical row r reads physical row rowidx[r]
  int aux;             // B operand, B_UNION_FLAT: number of pairs P (ld = pair stride K*49).  A operand with rowoff: column
                       // split of a GROUPED launch -- output columns >= aux read their rows through rowoff + M (a second
                       // gather table behind the first: subj_fc | obj_fc as one GEMM over stacked weights); 0 = none
  // Optional gather by ELEMENT OFFSET: logical row r (B_UNION_FLAT: pair r) starts at ptr + rowoff[r] floats.  This is
  // how a batch of clips is read where the caller left it (SttranInputs' per-clip pointer tables): the rows of one
  // operand then live in several allocations, `ptr` is the first clip's tensor and an offset may be any signed
  // distance inside the device's address space (pair_prep_kernel writes them).  Takes precedence over rowidx.
  const int64_t* rowoff;
  // rowidx operands: number of physical rows rowidx may address (0 = unknown).  gemm16_kernel (gemm_f32_t16.h) addresses
  // a GATHERED operand by 32-bit byte offsets from `ptr`: gemm_linear takes it only when span * ld * 4 < 4 GB is known.
  int64_t span;
};

// B-operand kinds.  B_KMAJOR: rows of W[N][K], K-contiguous (nn.Linear weights, im2col rows).
// B_CONV2: the 3x3 convolution of the mask branch (lib/sttran.py:342) as an implicit GEMM.  Row n of the B
// operand is an output position (pair, oy, ox), column k = (ky, kx, ci) (the weights are permuted to match
// when they are loaded); elements are gathered from the channel-last input [pair][7][7][128] on the fly (no
// im2col buffer).
// (The 7x7/2 convolution in front of it has its own kernel, kernels_maskconv.hip.)
// B_KMAJOR_PAD: K-major rows like B_KMAJOR, with the caller's guarantee that (1) every row of BOTH operands can be read
// up to the next multiple of 32 columns (finite values) and (2) B is zero there.  Then nothing has to be zeroed while
// staging: rows past M / N are loaded from a clamped row and never stored by the epilogue, the K tail multiplies
// whatever A holds by B's zeros.  This removes the select (4 v_cndmask per 16 bytes) from the main loop; it is the
// product path: weights are stored with zero-padded rows, and every activation buffer has a row stride of ceil32(K)
// floats whose pad columns are zeroed once and never written (api_layout.hip::ensure_workspace), so the K tail is
// 0 x 0 whatever earlier calls left in the workspace.  B_KMAJOR (with the select) takes any operands.
// B_UNION_FLAT: the 1x1 convolution union_func1 (lib/sttran.py:336,386) on the NCHW union_feat tensor U[P][K][49] read
// in place: GEMM column = pair * 49 + hw, running straight over the pair borders (N = 49 P exactly: no padded columns,
// all 256 output channels in ONE 256x128 tile, so U is fetched once).  A thread stages (column, 4 consecutive k): four
// coalesced dword loads (the lanes of a wave walk 64 consecutive columns = hw) and ONE ds_write_b128 into the ordinary
// K-major stage [column][k] -- the transposition happens in the registers, and the main loop (ds_read_b128 fragments)
// is the one of the nn.Linear path.  (Round 1's variant -- 128x256 tiles of five whole pairs, [k][hw] slabs copied flat
// into LDS and read back with ds_read_b32 -- fetched U twice and left 4.3 % of its columns empty: 116.5 -> 119.8 TFLOP/s.)
enum { B_KMAJOR = 0, B_CONV2 = 2, B_KMAJOR_PAD = 3, B_UNION_FLAT = 4 };
template <int BKIND> struct ConvGeo { static constexpr int KH = 1, S = 1, PAD = 0, HI = 1, HO = 1, CIN = 1, KREAL = 1; };
template <> struct ConvGeo<B_CONV2> {   // Conv2d(128, 256, kernel 3, padding 1) on 7x7 -> 7x7
  static constexpr int KH = 3, S = 1, PAD = 1, HI = 7, HO = 7, CIN = 128, KREAL = 1152;
};
constexpr int kUHW = 49;                                       // positions of a 7x7 union feature map

// ---- epilogues: called once per output element as epi(row, col, acc) --------------------
struct EpiLinear {
  float* C; int64_t ldc;
  const float* bias;        // [N] or null
  const float* rowbias;     // [2][rb_ld] extra bias selected by rowslot[out row] for col < rb_cols
  const uint8_t* rowslot;
  int rb_cols; int rb_ld;
  const float* scale;       // per-col affine applied after bias (BatchNorm eval), or null
  const float* shift;
  const float* res; int64_t ldres; const int* res_rowidx;   // residual add, optional gather
  int relu;
  const int* out_rowidx;    // optional scatter: GEMM row r is written to C row out_rowidx[r] (skipped if < 0)
  const int* out_rowidx2;   // optional second copy of the same row (skipped if < 0)
  __device__ __forceinline__ void put(int orow, int row, int col, float v) const {
    if (rowbias && col < rb_cols) v += rowbias[(int)rowslot[orow] * rb_ld + col];
    if (scale) v = v * scale[col] + shift[col];
    if (relu) v = relu_nan(v);
    if (res) v += res[(int64_t)(res_rowidx ? res_rowidx[row] : row) * ldres + col];
    C[(int64_t)orow * ldc + col] = v;
  }
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    if (bias) v += bias[col];
    if (!out_rowidx) { put(row, row, col, v); return; }
    const int o1 = out_rowidx[row];
    if (o1 >= 0) put(o1, row, col, v);
    if (out_rowidx2) {
      const int o2 = out_rowidx2[row];
      if (o2 >= 0) put(o2, row, col, v);
    }
  }
};

// relation heads (lib/sttran.py:404-409): cols [0,na) raw logits, the rest through sigmoid,
// written to three caller buffers.
struct EpiHeads {
  float* att; float* spa; float* con; const float* bias; int na, ns, nc;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    v += bias[col];
    if (col < na) { att[(int64_t)row * na + col] = v; return; }
    v = 1.f / (1.f + expf(-v));
    if (col < na + ns) spa[(int64_t)row * ns + (col - na)] = v;
    else con[(int64_t)row * nc + (col - na - ns)] = v;
  }
};

// conv3x3 as GEMM with M = out channel, N = (pair, hw): ReLU then eval-BatchNorm
// (lib/sttran.py:342-344: the BN follows the ReLU, so it cannot be folded into the conv),
// stored channel-major into V[p][c][hw] (the layout `.view(-1, 256*7*7)` flattens, :387).
struct EpiConvRelBn {
  float* V; const float* bias; const float* scale; const float* shift; int C; int HW;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    v = relu_nan(v + bias[row]) * scale[row] + shift[row];
    int p = col / HW, hw = col - p * HW;
    V[((int64_t)p * C + row) * HW + hw] = v;
  }
};

// union_func1 as GEMM (B_UNION_FLAT): row = out channel, col = pair * 49 + hw.
// V already holds the mask-conv branch: V[p][c][hw] += acc + bias[c]   (lib/sttran.py:386).
struct EpiUnionFlat {
  float* V; const float* bias; int C; int P;
  __device__ __forceinline__ float* at(int row, int col) const {
    const int p = col / kUHW, hw = col - p * kUHW;
    return V + ((int64_t)p * C + row) * kUHW + hw;
  }
  // `V += acc + bias` without a read-modify-write epilogue: the accumulators of the K range that starts a tile are
  // INITIALISED from V (kInit: 64 independent loads per thread, in flight together with the first operand loads), the
  // epilogue -- of the tile or, for a split tile, of the fix-up launch -- is a plain store.  Written as 64
  // `*dst += ...` the compiler has to keep every load behind the previous store (the addresses could alias): a chain of
  // round trips at the end of every tile that cost 7.5 % of the kernel (epilogue traffic ablated: 122 -> 132 TFLOP/s).
  static constexpr bool kInit = true;
  __device__ __forceinline__ bool init_on() const { return true; }
  __device__ __forceinline__ float init(int row, int col) const { return *at(row, col); }
  __device__ __forceinline__ void operator()(int row, int col, float v) const { *at(row, col) = v + bias[row]; }
};

// Vector epilogue of EpiLinear: four consecutive columns of one row (col % 4 == 0, col + 3 < N, every pointer
// 16-byte aligned at such columns -- the launcher checks).
struct EpiLinearV {
  static constexpr bool kVector = true;
  EpiLinear e;
  __device__ __forceinline__ void put4(int orow, int row, int col, f32x4 v) const {
    if (e.rowbias && col < e.rb_cols)
      v += *reinterpret_cast<const f32x4*>(e.rowbias + (int)e.rowslot[orow] * e.rb_ld + col);
    if (e.scale) v = v * *reinterpret_cast<const f32x4*>(e.scale + col) + *reinterpret_cast<const f32x4*>(e.shift + col);
    if (e.relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = relu_nan(v[c]);
    }
    if (e.res) v += *reinterpret_cast<const f32x4*>(e.res + (int64_t)(e.res_rowidx ? e.res_rowidx[row] : row) * e.ldres + col);
    *reinterpret_cast<f32x4*>(e.C + (int64_t)orow * e.ldc + col) = v;
  }
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
    if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + col);
    if (!e.out_rowidx) { put4(row, row, col, v); return; }
    const int o1 = e.out_rowidx[row];
    if (o1 >= 0) put4(o1, row, col, v);
    if (e.out_rowidx2) {
      const int o2 = e.out_rowidx2[row];
      if (o2 >= 0) put4(o2, row, col, v);
    }
  }
  __device__ __forceinline__ void operator()(int row, int col, float v) const { e(row, col, v); }
};
// The same epilogue for a whole strip of NV column groups of ONE output row, in two phases: every load the strip needs
// (bias, per-slot position bias, residual) is issued first, then the arithmetic and the stores.  Written as NV calls of
// vec() the compiler must keep the loads of group j + 1 behind the store of group j (C may alias what is loaded), which
// turns a tile's epilogue into a chain of NV dependent memory round trips per row (~0.5 us each: ~20 us per 128 x 176
// tile, measured as 4.4 K-steps of 61).  Same operation order as put4 (bit-identical results).
// cols[j] = first column of group j (col % 4 == 0); no per-column affine here (launchers route those to vec()).
// HRB / H2 / HRS (compile time) = per-slot position bias / second output row / residual present: the launch-uniform
// flags of EpiLinear as template parameters, so that the registers of an absent operand are never allocated
template <int NV, bool HRB, bool H2, bool HRS>
__device__ __forceinline__ void epi_linear_strip_t(const EpiLinear& e, int row, const int (&cols)[NV], const f32x4 (&acc)[NV]) {
  // loads never sit under a divergent condition: a disabled output row reads row 0 / slot 0 and is masked at the store
  const int o1 = e.out_rowidx ? e.out_rowidx[row] : row;
  const int o2 = H2 ? e.out_rowidx2[row] : -1;
  const float* rb1 = HRB ? e.rowbias + (int)e.rowslot[o1 >= 0 ? o1 : 0] * e.rb_ld : nullptr;
  const float* rb2 = (HRB && H2) ? e.rowbias + (int)e.rowslot[o2 >= 0 ? o2 : 0] * e.rb_ld : nullptr;
  const float* rs = HRS ? e.res + (int64_t)(e.res_rowidx ? e.res_rowidx[row] : row) * e.ldres : nullptr;
  const int rbmax = e.rb_cols - 4;                 // columns past rb_cols read (and discard) the last valid group
  const bool hb = e.bias != nullptr;
  f32x4 v[NV], r1[HRB ? NV : 1], r2[(HRB && H2) ? NV : 1], rr[HRS ? NV : 1];
  if (hb) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(e.bias + cols[j]);
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (HRB) {
#pragma unroll
    for (int j = 0; j < NV; ++j) r1[j] = *reinterpret_cast<const f32x4*>(rb1 + min(cols[j], rbmax));
    if constexpr (H2) {
#pragma unroll
      for (int j = 0; j < NV; ++j) r2[j] = *reinterpret_cast<const f32x4*>(rb2 + min(cols[j], rbmax));
    }
  }
  if constexpr (HRS) {
#pragma unroll
    for (int j = 0; j < NV; ++j) rr[j] = *reinterpret_cast<const f32x4*>(rs + cols[j]);
  }
  // v = acc + bias (an absent bias adds +0: x + 0 == x for every x the epilogue can see, -0 included as -0 + 0 = +0 only
  // for an exact -0 accumulator, which compares equal)
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = hb ? acc[j] + v[j] : acc[j];
  auto finish = [&](int orow, const f32x4* rbv) {
    if (orow < 0) return;
    float* dst = e.C + (int64_t)orow * e.ldc;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      f32x4 w = v[j];
      if constexpr (HRB) {
        const f32x4 w2 = w + rbv[j];
        const bool in = cols[j] < e.rb_cols;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = in ? w2[c] : w[c];
      }
      if (e.relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = relu_nan(w[c]);
      }
      if constexpr (HRS) w += rr[j];
      *reinterpret_cast<f32x4*>(dst + cols[j]) = w;
    }
  };
  finish(o1, r1);
  if constexpr (H2) finish(o2, r2);
}

template <int NV>
__device__ __forceinline__ void epi_linear_strip(const EpiLinear& e, int row, const int (&cols)[NV], const f32x4 (&acc)[NV]) {
  const bool hrb = e.rowbias != nullptr, h2 = e.out_rowidx2 != nullptr, hrs = e.res != nullptr;   // launch-uniform
  if (!hrb && !hrs) epi_linear_strip_t<NV, false, false, false>(e, row, cols, acc);
  else if (!hrb) epi_linear_strip_t<NV, false, false, true>(e, row, cols, acc);
  else if (!h2 && !hrs) epi_linear_strip_t<NV, true, false, false>(e, row, cols, acc);
  else if (h2 && !hrs) epi_linear_strip_t<NV, true, true, false>(e, row, cols, acc);
  else if (!h2) epi_linear_strip_t<NV, true, false, true>(e, row, cols, acc);
  else epi_linear_strip_t<NV, true, true, true>(e, row, cols, acc);
}

// Both rows of a wave's tile in ONE load -> store phase, for the two epilogue forms of the big launches: bias + residual
// (out-proj, FFN2) and bias + per-slot position bias (QKV of the middle decoder layers, encoder QKV with neither).  The
// bias strip is shared by the two rows, so the operands are 3 NV vectors next to the 2 NV accumulators.  Returns false
// (nothing done) for the forms it does not cover: output scatter (out_rowidx / out_rowidx2), residual AND position bias.
// NA = column groups of the accumulator array, [J0, J0 + NV) = the groups this call handles (group j starts at column
// col0 + 16 j): the 176-column tile runs it on two halves of its row so that the operand set stays below the registers
// the kernel has left next to 88 accumulator registers (one more load -> store phase per tile, no scratch traffic).
template <int NA, int J0, int NV>
__device__ __forceinline__ bool epi_linear_rows2(const EpiLinear& e, const int (&rows)[2], const bool (&valid)[2],
                                                 int col0, const f32x4 (&acc)[2][NA]) {
  const bool hrb = e.rowbias != nullptr, hrs = e.res != nullptr;
  if (e.out_rowidx || e.out_rowidx2 || (hrb && hrs) || e.scale) return false;
  const bool hb = e.bias != nullptr;
  f32x4 b[NV], x[2][NV];
  const float* src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // a row past M is not stored; its loads go to a row that exists (the lane's other row, else row 0 of the matrix)
    const int r = valid[i] ? rows[i] : (valid[0] ? rows[0] : 0);
    src[i] = hrs ? e.res + (int64_t)(e.res_rowidx ? e.res_rowidx[r] : r) * e.ldres
                 : hrb ? e.rowbias + (int)e.rowslot[r] * e.rb_ld : nullptr;
  }
  const int rbmax = e.rb_cols - 4;
  const int c0 = col0 + 16 * J0;
  if (hb) {
#pragma unroll
    for (int j = 0; j < NV; ++j) b[j] = *reinterpret_cast<const f32x4*>(e.bias + c0 + 16 * j);
  }
  if (hrs || hrb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NV; ++j) x[i][j] = *reinterpret_cast<const f32x4*>(src[i] + (hrb ? min(c0 + 16 * j, rbmax) : c0 + 16 * j));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!valid[i]) continue;
    float* dst = e.C + (int64_t)rows[i] * e.ldc;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      f32x4 w = hb ? acc[i][J0 + j] + b[j] : acc[i][J0 + j];
      if (hrb) {
        const f32x4 w2 = w + x[i][j];
        const bool in = c0 + 16 * j < e.rb_cols;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = in ? w2[c] : w[c];
      }
      if (e.relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = relu_nan(w[c]);
      }
      if (hrs) w += x[i][j];
      *reinterpret_cast<f32x4*>(dst + c0 + 16 * j) = w;
    }
  }
  return true;
}

// element-wise functors (heads, unaligned outputs) get a vec() that falls back to four scalar calls
template <class Epi>
struct EpiScalar4 {
  static constexpr bool kVector = true;
  Epi e;
  __device__ __forceinline__ void vec(int row, int col, f32x4 v) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) e(row, col + c, v[c]);
  }
  __device__ __forceinline__ void operator()(int row, int col, float v) const { e(row, col, v); }
};

// An epilogue with a vec() member asks for the SWAPPED MFMA operand ports (weights on the "A" port, activations on the
// "B" port): a lane's 16 accumulator registers then hold, for ONE output row (lane & 31), four groups of four
// CONSECUTIVE columns (8 q + 4 (lane >> 5) + {0..3}), so bias / residual are loaded and C is stored as 16-byte vectors
// -- a quarter of the memory instructions of the column-per-lane layout.  The k order inside an accumulator is the same
// either way (the result is bit-identical).
template <class Epi, class = void> struct EpiInit { static constexpr bool value = false; };
template <class Epi> struct EpiInit<Epi, decltype((void)Epi::kInit)> { static constexpr bool value = Epi::kInit; };
template <class Epi, class = void> struct EpiTraits { static constexpr bool swap = false; };
template <class Epi> struct EpiTraits<Epi, decltype((void)Epi::kVector)> { static constexpr bool swap = Epi::kVector; };

template <int BM_, int BN_, int WM_, int WN_, int BKIND_ = B_KMAJOR>
struct GemmTile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BKIND = BKIND_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int AV = BM * 8 / NT;                               // float4 loads per thread per step
  static constexpr int BV = BN * 8 / NT;
  static constexpr int STAGE_B = BN * kLdsStride;
  static constexpr int STAGE = BM * kLdsStride + STAGE_B;              // floats per LDS stage
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int GROUP_N = 8;                                   // tile_origin: N-tiles per group
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
  static_assert((BM * 8) % NT == 0, "A staging must divide evenly");
  static_assert((BN * 8) % NT == 0, "B staging must divide evenly");
  static_assert(BKIND != B_UNION_FLAT || (NT % BN == 0 && (NT / BN) * BV == 8), "B_UNION_FLAT: (column, k-group) per thread");
};

// XCD-aware, bijective remap of a linear block id: ids that are equal mod 8 share an XCD, so
// give each XCD a contiguous chunk of the logical tile order (guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// Tile index -> (M-tile, N-tile).  Consecutive indices sweep GN N-tiles of one M-tile, then the next M-tile, so the
// 32 workgroups of an XCD (consecutive indices, see xcd_remap) work on a compact ~4 x 8 block of tiles whose A and
// B panels are shared through that XCD's L2, instead of one long column of M-tiles that streams all of A for every
// N panel.  GN = 1 keeps the M-fastest order.
template <int GN>
__device__ __forceinline__ void tile_origin(int tile, int tiles_m, int tiles_n, int& tm, int& tn) {
  if (GN <= 1) { tm = tile % tiles_m; tn = tile / tiles_m; return; }
  const int per_group = tiles_m * GN;
  const int g = tile / per_group, local = tile - g * per_group;
  const int gn = min(GN, tiles_n - g * GN);
  tm = local / gn;
  tn = g * GN + (local - tm * gn);
}

// the same with the group width as a run-time value (the 16x16x4 kernels: tuned per launch shape by the launcher)
__device__ __forceinline__ void tile_origin_rt(int tile, int tiles_m, int tiles_n, int gn_max, int& tm, int& tn) {
  const int per_group = tiles_m * gn_max;
  const int g = tile / per_group, local = tile - g * per_group;
  const int gn = min(gn_max, tiles_n - g * gn_max);
  tm = local / gn;
  tn = g * gn_max + (local - tm * gn);
}

// ---- stream-K schedule ------------------------------------------------------------------------
// The iteration space (output tile, K-step) is cut into gridDim.x equal contiguous ranges, one per
// persistent workgroup, so every CU does the same number of MFMA steps whatever the tile count
// (the plain one-tile-per-workgroup grid loses up to half the chip to wave quantisation at the
// sizes of this path: e.g. 288 tiles of 128x128 for [2240,1936]x[1936,1936] on 256 CUs x 2 workgroups).
// A workgroup that covers a tile's whole K range runs the epilogue itself; a partial range is
// parked as raw accumulators in `slab` (at most two per workgroup: its first and its last tile) and
// summed, in fixed workgroup order (deterministic), by gemm_fixup_kernel, which then runs the same
// epilogue.  No inter-workgroup communication inside a launch.
struct SkRange {
  int begin, end;
};
// range of workgroup b out of G over `total` iterations: the first (total % G) workgroups get one
// iteration more.  base = total / G and rem = total % G come from the host (no device division).
__host__ __device__ __forceinline__ SkRange sk_range(int b, int base, int rem) {
  const int lo = b * base + (b < rem ? b : rem);
  return SkRange{lo, lo + base + (b < rem ? 1 : 0)};
}
// workgroup that owns iteration `it`
__host__ __device__ __forceinline__ int sk_owner(int it, int base, int rem) {
  const int big = rem * (base + 1);
  return it < big ? it / (base + 1) : rem + (it - big) / base;
}

template <class T, class Epi, int PIPE>
__global__ void __launch_bounds__(T::NT)
gemm_sk_kernel(GemmOperand A, GemmOperand B, int M, int N, int K, int tiles_m, int tiles, int ksteps, int dp_per_wg,
               int g_sk, int sk_base, int sk_rem, int half, float* __restrict__ slab, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN, AV = T::AV, BV = T::BV;
  constexpr bool CONV = T::BKIND == B_CONV2;
  constexpr bool PADDED = T::BKIND == B_KMAJOR_PAD;
  constexpr bool UFLAT = T::BKIND == B_UNION_FLAT;
  constexpr bool SWAP = EpiTraits<Epi>::swap;
  using Geo = ConvGeo<T::BKIND>;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int fr = lane & 31, fh = lane >> 5;
  const int kq4 = (tid & 7) * 4;
  const int a_off = (wm * (BM / T::WM) + fr) * kLdsStride + fh * 4;
  // B fragment base offsets per 32-column MFMA tile of this wave
  int b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / T::WN) + j * 32 + fr;
    b_off[j] = (BM + col) * kLdsStride + fh * 4;
  }

  // Hybrid schedule: every workgroup first runs dp_per_wg WHOLE tiles (data-parallel, nothing parked),
  // then the leftover tiles (fewer than one per workgroup) are stream-K'd over g_sk workgroups.
  const int G = gridDim.x;
  const int blk = xcd_remap(blockIdx.x, G);      // neighbouring ranges (shared weight panels) on one XCD
  const int tiles_dp = dp_per_wg * G;
  const SkRange rg = blk < g_sk ? sk_range(blk, sk_base, sk_rem) : SkRange{0, 0};
  // Workgroups dispatched after the first one-per-CU wave (blockIdx.x >= half) walk their work in the opposite order
  // -- stream-K range first, whole tiles after -- so that the workgroups sharing a CU reach their epilogues (no MFMAs)
  // at different times and one keeps the matrix pipe busy while the other stores.  Speed only: any order is correct.
  const bool sk_first = (int)blockIdx.x >= half;

  int dp_done = 0;
  for (int it = rg.begin; dp_done < dp_per_wg || it < rg.end;) {
    int tile, ks0, ks1;
    const bool dp = sk_first ? !(it < rg.end) : dp_done < dp_per_wg;
    if (dp) {
      tile = dp_done * G + blk;                   // at any moment the workgroups of an XCD hold consecutive tiles
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
    const int k_begin = ks0 * kBK;
    const int k_end = min(K, ks1 * kBK);

    // Staging slots.  Rows past M/N and the K tail are loaded from a valid address (row 0 / k 0) and
    // zeroed by a select at the LDS write, so the loads stay unconditional (no exec-mask branches in
    // the loop) and the wait for them sits behind the MFMAs of the current step.
    const float* pa[AV]; const float* pb[BV];
    bool va[AV], vb[BV];
    int sb[BV];                                    // LDS float offset of each B piece
    int cy[BV], cx[BV], cm[BV];                    // B_CONV: input row/col origin of the output position; element mask
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int g = m0 + (tid >> 3) + i * (NT >> 3);
      va[i] = g < M;
      const int64_t* ro = A.rowoff;
      if (ro && A.aux > 0 && n0 >= A.aux) ro += M;           // grouped launch: the second column group's gather table
      pa[i] = A.ptr + (ro ? (va[i] ? ro[g] : (int64_t)0)
                          : (int64_t)(va[i] ? (A.rowidx ? A.rowidx[g] : g) : 0) * A.ld) + kq4;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      if constexpr (UFLAT) {
        // thread = (column tid % BN, k-group (tid / BN) * BV + i): columns past N read pair 0 (dropped by the epilogue)
        const int col = tid % BN, kg = (tid / BN) * BV + i;
        const int n = n0 + col;
        vb[i] = n < N;
        const int nn = vb[i] ? n : 0, p = nn / kUHW, hw = nn - p * kUHW;
        sb[i] = (BM + col) * kLdsStride + kg * 4;
        pb[i] = B.ptr + (B.rowoff ? B.rowoff[p] : (int64_t)p * B.ld) + kg * 4 * kUHW + hw;
      } else if constexpr (CONV) {
        const int g = n0 + (tid >> 3) + i * (NT >> 3);
        vb[i] = g < N;
        const int gg = vb[i] ? g : 0;
        const int pr = gg / (Geo::HO * Geo::HO), pos = gg - pr * (Geo::HO * Geo::HO);
        const int oy = pos / Geo::HO, ox = pos - oy * Geo::HO;
        cy[i] = oy * Geo::S - Geo::PAD;
        cx[i] = ox * Geo::S - Geo::PAD;
        cm[i] = 0;
        sb[i] = (BM + (tid >> 3) + i * (NT >> 3)) * kLdsStride + kq4;
        pb[i] = B.ptr + (int64_t)pr * (Geo::CIN * Geo::HI * Geo::HI);
      } else {
        const int g = n0 + (tid >> 3) + i * (NT >> 3);
        vb[i] = g < N;
        sb[i] = (BM + (tid >> 3) + i * (NT >> 3)) * kLdsStride + kq4;
        pb[i] = B.ptr + (int64_t)(vb[i] ? (B.rowidx ? B.rowidx[g] : g) : 0) * B.ld + kq4;
      }
    }
    (void)cy; (void)cx; (void)cm;
    f32x4 ra[AV], rb[BV];
    bool kok_a = true, kok_b = true;               // validity of the K range currently held in ra / rb
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int ka = 0, kb_src = 0;                        // element offsets of the next load (A side, B side)
    auto set_k = [&](int k0) {
      if constexpr (PADDED) {
        // whole K-steps are loaded as they are (B is zero past K); only the dummy step after the range is redirected
        ka = k0 < ks1 * kBK ? k0 : 0;
        kb_src = ka;
      } else {
        // K-major rows: a float4 is all-in or all-out of [k_begin,k_end) because K % 4 == 0.
        // The union conv needs K % 32 == 0 (checked by the launcher): a K-step is never partial.
        kok_a = (k0 + kq4) < k_end;
        kok_b = UFLAT ? k0 < k_end : kok_a;
        ka = kok_a ? k0 : -kq4;                    // out of range: the row's first 16 bytes (pa / pb carry + kq4; K may be < 32)
        kb_src = UFLAT ? (kok_b ? k0 * kUHW : 0) : (CONV ? k0 + kq4 : ka);
      }
    };
    auto load_piece = [&](int n) {
      if (n < AV) {
        ra[n] = *reinterpret_cast<const f32x4*>(pa[n] + ka);
      } else if constexpr (CONV) {
        // K is ordered (ky, kx, ci) for this conv (weights permuted to match at load time) and the input is
        // channel-last [pair][iy][ix][ci], so the four k of a piece are four consecutive input channels at ONE
        // tap = one 16-byte load; the tap and its bounds test are wave-uniform per K-step (128 channels = 4
        // K-steps per tap)
        const int i = n - AV;
        const int k0 = kb_src - kq4;                       // uniform
        const int tap = k0 / Geo::CIN, ky = tap / Geo::KH, kx = tap - ky * Geo::KH;
        const int ci = k0 - tap * Geo::CIN + kq4;
        const int iy = cy[i] + ky, ix = cx[i] + kx;
        const bool ok = k0 < k_end && (unsigned)iy < (unsigned)Geo::HI && (unsigned)ix < (unsigned)Geo::HI;
        rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + (ok ? (iy * Geo::HI + ix) * Geo::CIN + ci : 0));
        cm[i] = ok ? 0xF : 0;
      } else if constexpr (UFLAT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[n - AV][e] = pb[n - AV][kb_src + e * kUHW];
      } else {
        rb[n - AV] = *reinterpret_cast<const f32x4*>(pb[n - AV] + kb_src);
      }
    };
    auto store_piece = [&](int n, float* stage) {
      if (n < AV) {
        f32x4* dst = reinterpret_cast<f32x4*>(stage + ((tid >> 3) + n * (NT >> 3)) * kLdsStride + kq4);
        // the union conv and the 3x3 conv have M = 256 output channels (a whole number of tiles) and K % 32 == 0
        // (their launchers check both): every A piece is valid, like on the padded path
        if constexpr (PADDED || CONV || UFLAT) *dst = ra[n];
        else *dst = (va[n] && kok_a) ? ra[n] : zero4;
      } else {
        const int i = n - AV;
        if constexpr (CONV) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (vb[i] && ((cm[i] >> e) & 1)) ? rb[i][e] : 0.f;
          *reinterpret_cast<f32x4*>(stage + sb[i]) = v;
        } else if constexpr (PADDED) {
          *reinterpret_cast<f32x4*>(stage + sb[i]) = rb[i];
        } else if constexpr (UFLAT) {
          // columns past N were loaded from pair 0 (clamped): they are never stored by the epilogue, nothing to zero
          *reinterpret_cast<f32x4*>(stage + sb[i]) = rb[i];
        } else {
          *reinterpret_cast<f32x4*>(stage + sb[i]) = (vb[i] && kok_b) ? rb[i] : zero4;
        }
      }
    };
    // fragment group kb of a stage: k = kb*8 + 4*(lane>>5) + {0..3} for A rows and B alike
    auto read_frags = [&](const float* stage, int kb, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const f32x4*>(stage + a_off + i * 32 * kLdsStride + kb * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fb[j] = *reinterpret_cast<const f32x4*>(stage + b_off[j] + kb * 8);
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if constexpr (EpiInit<Epi>::value && !SWAP) {
      // C += A B: the K range that starts a tile accumulates onto the output's old values (see EpiUnionFlat)
      if (ks0 == 0 && epi.init_on()) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / T::WN) + j * 32 + fr;
            const int rbase = m0 + wm * (BM / T::WM) + i * 32 + 4 * fh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = rbase + (e & 3) + 8 * (e >> 2);
              if (row < M && col < N) acc[i][j][e] = epi.init(row, col);
            }
          }
      }
    }
    // one MFMA; SWAP feeds the B fragment to the "A" port (see EpiTraits)
    auto mfma1 = [&](f32x16& c, float a, float b) {
      if constexpr (SWAP) c = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c, 0, 0, 0);
      else c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    };
    auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma1(acc[i][j], fa[i][e], fb[j][e]);
    };

    // Operand kinds without per-step validity state (padded K-major rows, union columns, conv taps) prefetch TWO K-steps
    // ahead: the global loads of step t+2 are issued during step t into one of two register sets and written to
    // LDS during step t+1, so a load has a K-step and a half (~5 us) to arrive instead of half a K-step -- what
    // the operands need when they come from HBM rather than from a warm cache (tools/gemm_bench.py --cold).
    constexpr bool DEEP = (PADDED || CONV || UFLAT) && PIPE == 1;
    // load slots of the two-deep pipeline: a B_UNION_FLAT piece is four dword loads, each its own slot
    constexpr int NPL = AV + (UFLAT ? 4 * BV : BV);
    f32x4 ra2[DEEP ? AV : 1], rb2[DEEP ? BV : 1];       // second register set
    int cm2[BV];                                         // ... and its conv tap masks
    (void)cm2;
    // element offsets of K-step u of this range (the steps past its end read step 0 again, into an idle buffer)
    auto koff_a = [&](int u) { const int k0 = k_begin + u * kBK; return k0 < ks1 * kBK ? k0 : 0; };
    auto koff_b = [&](int u) { return UFLAT ? koff_a(u) * kUHW : koff_a(u); };
    auto load_to = [&](int n, auto& RA, auto& RB, auto& CM, int ka_, int kb_) {
      if (n < AV) {
        RA[n] = *reinterpret_cast<const f32x4*>(pa[n] + ka_);
      } else if constexpr (CONV) {
        // (ky, kx, ci) K order over a channel-last input: one tap per K-step, 4 channels = one 16-byte load; a tap
        // outside the image is a zero piece (CM), loaded from a clamped address
        const int i = n - AV;
        const int tap = ka_ / Geo::CIN, ky = tap / Geo::KH, kx = tap - ky * Geo::KH;
        const int ci = ka_ - tap * Geo::CIN + kq4;
        const int iy = cy[i] + ky, ix = cx[i] + kx;
        const bool ok = (unsigned)iy < (unsigned)Geo::HI && (unsigned)ix < (unsigned)Geo::HI;
        RB[i] = *reinterpret_cast<const f32x4*>(pb[i] + (ok ? (iy * Geo::HI + ix) * Geo::CIN + ci : 0));
        CM[i] = ok;
      } else if constexpr (UFLAT) {
        const int i = (n - AV) >> 2, e = (n - AV) & 3;       // slot = (piece, k inside the piece)
        RB[i][e] = pb[i][kb_ + e * kUHW];
      } else {
        RB[n - AV] = *reinterpret_cast<const f32x4*>(pb[n - AV] + kb_);
      }
    };
    auto store_from = [&](int n, float* stage, const auto& RA, const auto& RB, const auto& CM) {
      if (n < AV) {
        *reinterpret_cast<f32x4*>(stage + ((tid >> 3) + n * (NT >> 3)) * kLdsStride + kq4) = RA[n];
      } else {
        const int i = n - AV;
        if constexpr (CONV) *reinterpret_cast<f32x4*>(stage + sb[i]) = CM[i] ? RB[i] : zero4;
        else *reinterpret_cast<f32x4*>(stage + sb[i]) = RB[i];
      }
    };
    if constexpr (DEEP) {
#pragma unroll
      for (int n = 0; n < NPL; ++n) load_to(n, ra, rb, cm, koff_a(0), koff_b(0));
#pragma unroll
      for (int n = 0; n < NPL; ++n) load_to(n, ra2, rb2, cm2, koff_a(1), koff_b(1));
#pragma unroll
      for (int n = 0; n < AV + BV; ++n) store_from(n, smem, ra, rb, cm);
    } else {
      set_k(k_begin);
#pragma unroll
      for (int n = 0; n < AV + BV; ++n) load_piece(n);
#pragma unroll
      for (int n = 0; n < AV + BV; ++n) store_piece(n, smem);
    }
    __syncthreads();
    if constexpr (PIPE == 0) {
      // plain loop (kept for A/B runs of tools/gemm_bench.py): fragments of group kb are read right
      // before its MFMAs, staging clustered before/after the MFMAs, one barrier per K-step
      f32x4 fa[TM], fb[TN];
      for (int t = 0; t < nsteps; ++t) {
        const float* cur = smem + (t & 1) * T::STAGE;
        set_k(k_begin + (t + 1) * kBK);
#pragma unroll
        for (int n = 0; n < AV + BV; ++n) load_piece(n);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          read_frags(cur, kb, fa, fb);
          mma(fa, fb);
        }
#pragma unroll
        for (int n = 0; n < AV + BV; ++n) store_piece(n, smem + ((t + 1) & 1) * T::STAGE);
        __syncthreads();
      }
    } else {
      // Software pipeline: fragment group kb+1 is read from LDS while the MFMAs of group kb run, and
      // the last group of a K-step is held in registers across the barrier, so its 4*TM*TN MFMAs
      // cover the barrier wait and the LDS latency of the next step's first fragment read.
      // The staging work is spread one piece per MFMA gap (an MFMA occupies the matrix pipe for 64
      // cycles but the issue port for 8): group 0 carries the global loads of the next tile, group 2
      // carries their select + ds_write, so neither costs issue time of its own.  Loads and stores
      // run on the last step too (from a clamped address, into the idle buffer): no branches.
      // sched_barrier / sched_group_barrier pin this order: left alone, hipcc sinks every load to
      // just before its use, which exposes the full memory latency each step.
      constexpr int NP = AV + BV, NM = 4 * TM * TN;
      f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
      read_frags(smem, 0, fa0, fb0);
      // one K-step of the two-deep variant: `RAL/RBL` receive the loads of step t+2, `RAS/RBS` (loaded one step
      // earlier) are written to the other LDS stage
      auto deep_step = [&](int t, auto& RAL, auto& RBL, auto& CML, const auto& RAS, const auto& RBS, const auto& CMS) {
        const float* cur = smem + (t & 1) * T::STAGE;
        float* nxt = smem + ((t + 1) & 1) * T::STAGE;
        const int ka_ = koff_a(t + 2), kb_ = koff_b(t + 2);
        read_frags(cur, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        {
          int n = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                mfma1(acc[i][j], fa0[i][e], fb0[j][e]);
                if (n < NPL) load_to(n, RAL, RBL, CML, ka_, kb_);
                ++n;
              }
#pragma unroll
          for (; n < NPL; ++n) load_to(n, RAL, RBL, CML, ka_, kb_);
#pragma unroll
          for (int q = 0; q < NPL && q < NM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur, 2, fa0, fb0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur, 3, fa1, fb1);
        {
          int n = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                mfma1(acc[i][j], fa0[i][e], fb0[j][e]);
                if (n < NP) store_from(n, nxt, RAS, RBS, CMS);
                ++n;
              }
#pragma unroll
          for (; n < NP; ++n) store_from(n, nxt, RAS, RBS, CMS);
#pragma unroll
          for (int q = 0; q < NP && q < NM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
            if constexpr (CONV) __builtin_amdgcn_sched_group_barrier(0x2, 4, 1);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(nxt, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
      };
      if constexpr (DEEP) {
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {
          deep_step(t, ra, rb, cm, ra2, rb2, cm2);          // even step: set 0 is free (stored), set 1 holds step t+1
          deep_step(t + 1, ra2, rb2, cm2, ra, rb, cm);
        }
        if (t < nsteps) deep_step(t, ra, rb, cm, ra2, rb2, cm2);
      } else
      for (int t = 0; t < nsteps; ++t) {
        const float* cur = smem + (t & 1) * T::STAGE;
        float* nxt = smem + ((t + 1) & 1) * T::STAGE;
        set_k(k_begin + (t + 1) * kBK);
        read_frags(cur, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        {
          int n = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                mfma1(acc[i][j], fa0[i][e], fb0[j][e]);
                if (n < NP && PIPE != 4 && PIPE != 5) load_piece(n);
                ++n;
              }
#pragma unroll
          for (; n < NP; ++n) if (PIPE != 4 && PIPE != 5) load_piece(n);       // tiles with fewer MFMAs per group than pieces
          // one global load per MFMA gap (sched_group_barrier masks: 0x8 MFMA, 0x20 VMEM read)
#pragma unroll
          for (int q = 0; q < NP && q < NM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur, 2, fa0, fb0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur, 3, fa1, fb1);
        {
          int n = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                mfma1(acc[i][j], fa0[i][e], fb0[j][e]);
                if (n < NP && PIPE != 4 && PIPE != 5) store_piece(n, nxt);
                ++n;
              }
#pragma unroll
          for (; n < NP; ++n) if (PIPE != 4 && PIPE != 5) store_piece(n, nxt);
          // one select + ds_write per MFMA gap (0x2 VALU, 0x200 DS write)
#pragma unroll
          for (int q = 0; q < NP && q < NM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 1);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PIPE != 3 && PIPE != 5) __syncthreads();     // PIPE 3/4/5: timing-only ablations (wrong results)
        read_frags(nxt, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);      // issue these reads BEFORE the held-over MFMA group, which then hides them
        mma(fa1, fb1);
      }
    }

    if constexpr (SWAP) {
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
        // partial K range: park the raw accumulators as 16-byte vectors, register-major (1 KB per wave-instruction);
        // gemm_fixup_vec_kernel sums them.  (Reducing them inside the launch -- write-through parking, an arrival
        // counter per tile, the last arriver sums -- was built and measured: correct, but slower than this fix-up
        // launch at every size)
        f32x4* sp = reinterpret_cast<f32x4*>(slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN)) + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              sp[((i * TN + j) * 4 + q) * NT] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      }
    } else
    if (nsteps == ksteps) {
      // ---- whole tile: epilogue.  C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * (BN / T::WN) + j * 32 + fr;
          const int rbase = m0 + wm * (BM / T::WM) + i * 32 + 4 * fh;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = rbase + (e & 3) + 8 * (e >> 2);
            if (row < M && col < N) epi(row, col, acc[i][j][e]);
          }
        }
      }
    } else {
      // ---- partial K range: park the raw accumulators (slot 0 = this workgroup's first tile,
      //      slot 1 = its last; only the stream-K region parks), register-major so that every store is 256 contiguous bytes per wave
      float* sp = slab + ((int64_t)blk * 2 + (it == rg.begin ? 0 : 1)) * (BM * BN) + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) sp[((i * TN + j) * 16 + e) * NT] = acc[i][j][e];
    }
    if (!dp) it += nsteps;
  }
}

// Sums the parked partial accumulators of every split tile in ascending workgroup order and runs
// the epilogue.  grid = (tiles, TM*TN): one workgroup per 32x32-per-wave register group of a tile, so
// the re-read of the parked data is spread over many CUs even when only a few tiles were split.
// Tiles computed whole by one workgroup return at once.
template <class T, class Epi>
__global__ void __launch_bounds__(T::NT)
gemm_fixup_kernel(int M, int N, int tiles_m, int tiles_n, int ksteps, int g_sk, int sk_base, int sk_rem, int tiles_dp,
                  const float* __restrict__ slab, Epi epi) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TN = T::TN;
  const int tile = blockIdx.x;                   // index inside the stream-K (leftover) region
  const int t0 = tile * ksteps, t1 = t0 + ksteps;
  const int b_lo = sk_owner(t0, sk_base, sk_rem), b_hi = sk_owner(t1 - 1, sk_base, sk_rem);
  if (b_lo == b_hi) return;                      // computed whole by one workgroup: nothing parked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN, fr = lane & 31, fh = lane >> 5;
  // blockIdx.y = (register group ij, quarter eq): each thread sums 4 of the 16 accumulator registers of
  // its lane, so the loads of up to eight contributors (32 per thread) are in flight together and the
  // grid is four times wider -- the kernel is a latency chain, not a bandwidth problem
  const int ij = blockIdx.y >> 2, eq = blockIdx.y & 3, i = ij / TN, j = ij % TN;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* base = slab + ((int64_t)ij * 16 + eq * 4) * NT + tid;
  // workgroup b parked this tile in slot 0 if the tile holds the start of b's range, else slot 1;
  // only b_lo can start before the tile.  The sum order (ascending b) is fixed: reproducible.
  for (int b = b_lo; b <= b_hi; b += 8) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = b + u;
      const bool ok = bb <= b_hi;
      const int slot = (bb == b_lo && sk_range(bb, sk_base, sk_rem).begin < t0) ? 1 : 0;
      const float* sp = base + ((int64_t)(ok ? bb : b_lo) * 2 + slot) * (BM * BN);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = ok ? sp[e * NT] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
  }
  const int gt = tiles_dp + tile;                // global tile index
  int tile_m, tile_n;
  tile_origin<T::GROUP_N>(gt, tiles_m, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int col = n0 + wn * (BN / T::WN) + j * 32 + fr;
  const int rbase = m0 + wm * (BM / T::WM) + i * 32 + 4 * fh + 8 * eq;   // e = 4*eq + r: row = r + 8*eq
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rbase + r;
    if (row < M && col < N) epi(row, col, acc[r]);
  }
}

// The same for the swapped-port (vector) accumulator layout: sums the parked 16-byte vectors of every split tile in
// ascending workgroup order and runs the vector epilogue.  // grid = (tiles_sk, TM * TN * 4): one workgroup per 16-byte register group of a tile.
template <class T, class Epi>
__global__ void __launch_bounds__(T::NT)
gemm_fixup_vec_kernel(int M, int N, int tiles_m, int tiles_n, int ksteps, int g_sk, int sk_base, int sk_rem, int tiles_dp,
                      const float* __restrict__ slab, Epi epi) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TN = T::TN;
  const int tile = blockIdx.x;
  const int t0 = tile * ksteps, t1 = t0 + ksteps;
  const int b_lo = sk_owner(t0, sk_base, sk_rem), b_hi = sk_owner(t1 - 1, sk_base, sk_rem);
  if (b_lo == b_hi) return;                      // computed whole by one workgroup: nothing parked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN, fr = lane & 31, fh = lane >> 5;
  const int ijq = blockIdx.y, q = ijq & 3, ij = ijq >> 2, i = ij / TN, j = ij % TN;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* base = reinterpret_cast<const f32x4*>(slab) + (int64_t)ijq * NT + tid;
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
  const int row = tile_m * BM + wm * (BM / T::WM) + i * 32 + fr;
  const int col = tile_n * BN + wn * (BN / T::WN) + j * 32 + 8 * q + 4 * fh;
  if (row < M) {
    if (col + 3 < N) epi.vec(row, col, acc);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c < N) epi(row, col + c, acc[c]);
    }
  }
}

}  // namespace sttran
