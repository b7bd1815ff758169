// kernels_maskconv.hip -- first half of the spatial-mask branch of the pair fusion in ONE kernel:
//   Conv2d(2, 128, kernel 7, stride 2, padding 3) -> ReLU -> BatchNorm2d(128, eval) -> MaxPool2d(3, 2, 1)
//   masks [P, 2, 27, 27] -> C2 [P, 7, 7, 128] (channel-last)              (lib/sttran.py:337-341)
// The 14x14x128 convolution output (100 KB per pair) never reaches HBM: four waves compute it for one pair in MFMA
// accumulators, apply bias / ReLU / BN, pool through a small LDS buffer and store the 7x7x128 result (25 KB per
// pair), channel-last: the layout the implicit-GEMM 3x3 convolution gathers best.
//
// Per pair the convolution is a GEMM  [128 channels] x [K = 2 x 49 taps] x [196 positions]:
//   * one wave owns 32 output channels and all 196 positions (7 MFMA 32x32 column blocks, 224 columns);
//   * its weights stay in registers for the whole launch (13 groups of 4 k per lane-half = 52 VGPRs):
//     K is ordered (tap, channel-of-the-lane-half), i.e. in every v_mfma_f32_32x32x2_f32 the lower 32
//     lanes carry input channel 0 and the upper 32 lanes input channel 1 of the same tap;
//   * the B operand is never materialised (no im2col): the pair's two 27x27 masks sit zero-padded to 33x33
//     in LDS, and B[k][n] is a single LDS read at  lane_base + column_block_offset + tap_offset, where both offsets
//     are compile-time immediates -- no address arithmetic and no bounds tests in the loop;
//   * LDS layout of a padded plane (round 6; rounds 1-5 kept it row-major and read it at stride 2 -- every operand read
//     a 2-way bank conflict, SQ_LDS_BANK_CONFLICT = 63 % of the kernel's LDS cycles together with the pooling reads):
//     the stride-2 convolution reads padded (2 oy + ky, 2 ox + kx), so the plane is split into its four (row parity,
//     column parity) sub-planes [17][17]; tap (ky, kx) reads sub-plane (ky & 1, kx & 1) at (oy + ky / 2, ox + kx / 2) --
//     neighbouring output columns are neighbouring floats.  A 32-column MFMA block holds TWO output rows (28 positions,
//     4 padding columns: 7 blocks = 14 rows = the same 224 columns as before): its lanes read 14 + 14 consecutive floats
//     17 apart (+ 4 padding lanes on the 4 banks left over) = 32 distinct banks of `ds_read_b32`'s 32.
//
// What bounds it (round 5, s_memtime traces + PMC): an fp32 MFMA executes on the SIMD's vector ALUs -- beside a wave that
// issues v_mfma_f32_32x32x2_f32 back to back the other wave of the SIMD gets almost no vector instruction through (an
// 8-wave ping-pong form of this kernel, one group in its MFMA phase while the other runs its epilogue, with and without
// s_setprio, measured SLOWER: the epilogue's first round took 24 k cycles beside the MFMAs, the other three 3 k each after
// them).  So a pair costs its SIMD  350 MFMAs x 64 cycles + every vector instruction of the epilogue x 4 cycles,  and only
// LDS / memory latencies overlap.  Hence: 350 instead of 364 MFMAs (tap 49 carries the BIAS: weight = bias, operand = 1.0;
// taps 50, 51 are not executed), a hand-pipelined K loop (LDS reads one unit ahead), and an epilogue with a third of the
// vector instructions it had (-inf bordered pooling planes: nine unconditional reads per window; v_max3 + unordered
// compares for the NaN-propagating maximum; results leave as whole 128-byte lines, 7 instead of 28 store instructions).
// Two independent 4-wave workgroups per CU (one wave of each per SIMD) cover each other's LDS and memory waits.
#include <algorithm>

#include "kernels.h"

#ifdef STTRAN_MC_TRACE
// experiment builds only (make EXTRA="-DSTTRAN_GEMM_EXPERIMENT -DSTTRAN_MC_TRACE"): s_memtime stamps of the first workgroups'
// half-periods -- [workgroup 0..3][group 0..1][half-period 0..31][start, work done, barrier passed]
__device__ unsigned long long g_mc_trace[4 * 2 * 32 * 3];
__device__ unsigned long long g_mc_fine[64];
#define MC_FINE(k) do { if (fine_on) g_mc_fine[k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int sttran_debug_mc_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mc_trace), sizeof(g_mc_trace));
}
extern "C" int sttran_debug_mc_fine(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mc_fine), sizeof(g_mc_fine));
}
#else
#define MC_FINE(k) do { } while (0)
#endif

namespace sttran {
namespace {

constexpr int kMcSubW = 17;                  // a parity sub-plane of the 33 x 33 padded plane (27 + 2 * 3): [17][17]
constexpr int kMcSub = kMcSubW * kMcSubW;    // 289 floats
constexpr int kMcPlane = 4 * kMcSub;         // 1156 floats per padded input channel: sub-planes (row parity, column parity)
constexpr int kMcMask = 2 * kMcPlane;        // one pair
constexpr int kMcGroups = 13;                // 52 taps (49 real) in groups of 4 (the register layout of the weights)
constexpr int kMcUnits = 25;                 // executed K: 25 units of two taps (tap 49 has zero weights; 50, 51 are skipped)
constexpr int kMcPoolCh = 8;                 // channels pooled per round and wave
constexpr int kMcPoolW = 15;                 // a pooled channel's 14 x 14 conv map with a -inf row above and column left of it
constexpr int kMcPoolPlane = kMcPoolW * kMcPoolW;   // 225 floats
// where the pooled channel c of a round starts in the wave's pooling buffer: 225 c + 12 for every two channels.  The pooling
// reads of a 32-lane group are 4 channels (0..3 or 4..7) x 7 window columns at stride 2; with these starts (0, 1, 14, 15 and
// 16, 17, 30, 31 mod 32) the four channels' seven addresses fall on 28 distinct banks (a plain 225 c gave starts 0, 1, 2, 3
// and channel-fastest lanes: up to 4-way conflicts on each of the 63 reads of a round)
__host__ __device__ constexpr int pool_base(int c) { return c * kMcPoolPlane + 12 * (c >> 1); }
constexpr int kMcPool = pool_base(kMcPoolCh - 1) + kMcPoolPlane;   // 1836 floats per wave
constexpr int kMcWaves = 4;
constexpr int kMcOutRow = 36;                // a wave's pooled result waits in LDS as [49 positions][32 channels], rows 36 floats apart
constexpr int kMcOut = 49 * kMcOutRow;       // (36 px + ch: the 28 lanes of a write hit 28 banks)
constexpr int kMcLdsFloats = 2 * kMcMask + kMcWaves * (kMcPool + kMcOut) + 3 * 128;

// float offset of tap t < 49 = (ky, kx) from the lane's base (oy, ox) of sub-plane (0, 0)
__host__ __device__ constexpr int tap_offset(int t) {
  return (((t / 7) & 1) * 2 + ((t % 7) & 1)) * kMcSub + ((t / 7) >> 1) * kMcSubW + ((t % 7) >> 1);
}

__global__ void __launch_bounds__(256, 2)
mask_conv1_pool_kernel(const float* __restrict__ masks, const int64_t* __restrict__ mask_off, const float* __restrict__ w0p,
                       const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
                       float* __restrict__ c2, int P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: scalar registers
  const int fr = lane & 31, fh = lane >> 5;
  float* pool = lds + 2 * kMcMask + wave * kMcPool;                      // wave-private 8 x [15][15], row 0 / column 0 = -inf
  float* outb = lds + 2 * kMcMask + kMcWaves * kMcPool + wave * kMcOut;   // wave-private [49][36]
  float* par = lds + 2 * kMcMask + kMcWaves * (kMcPool + kMcOut);         // bias | scale | shift, [3][128]

  // weights of this lane: channel 32*wave + fr, k = (group, half, e)  (w0p is [128][104] in that order)
  f32x4 a[kMcGroups];
#pragma unroll
  for (int g = 0; g < kMcGroups; ++g)
    a[g] = *reinterpret_cast<const f32x4*>(w0p + (wave * 32 + fr) * (kMcGroups * 8) + g * 8 + fh * 4);
  // Per-lane maps: ONE base register each, everything else is an immediate.
  //  * conv column (block j, lane fr): output position (oy, ox) = (2 j + hi, fr - 14 hi), hi = fr >= 14; fr >= 28 is padding.
  //    B base of the column in its lane-half's channel plane:  fh * 1156 + 17 oy + ox = lb0 + 34 j;  the padding lanes read
  //    (and discard) the floats on the four banks the 28 real lanes leave free: lb0 = 14, 15, 16, 31 past the block's start.
  //    Where its activation goes in the wave's pooling plane ((oy + 1, ox + 1) of the 15 x 15 bordered map): pw0 + 30 j.
  //  * pooling: one pooled ROW per iteration (py = i): lane = 32 g + l, l < 28 -> channel 4 g + l / 7, column px = l % 7.
  //    The 3x3 / stride-2 / padding-1 window of (py, px) covers conv rows 2py-1 .. 2py+1 and columns 2px-1 .. 2px+1; with
  //    the -inf border row / column in front of the map every window is nine unconditional reads at pb0 + 30 i + {0,1,2} +
  //    {0,15,30} (MaxPool2d pads with -inf, lib/sttran.py:341).  Lanes l >= 28 read lane 0's window (same addresses:
  //    broadcast) and store nothing.  The pooled value goes to the wave's LDS result block at ob0 + 252 i + 8 q; when the
  //    four rounds are done the block leaves as WHOLE 128-byte lines (a position's 32 channels of this wave), 16 bytes per
  //    lane: round 1-4's dword stores wrote a line in four 32-byte pieces, one per round, and the first piece of every
  //    line stalled the wave for the line's allocation (s_memtime trace: the first round's stores took 16 k cycles, the
  //    others 1.7 k).
  const int hi = fr >= 14 ? 1 : 0, oxl = fr - 14 * hi;
  const int lb0 = fh * kMcPlane + (fr < 28 ? kMcSubW * hi + oxl : (fr == 31 ? 31 : fr - 14));
  const int pw0 = (kMcPoolW + 1) + kMcPoolW * hi + oxl + fh * pool_base(4);
  const int pl = lane & 31, pact = pl < 28, pch = pact ? pl / 7 : 0, ppx = pact ? pl % 7 : 0;
  const int pb0 = (lane >> 5) * pool_base(4) + pch * kMcPoolPlane + 12 * (pch >> 1) + 2 * ppx;     // pool_base(4 g + pch) + 2 px
  const int ob0 = ppx * kMcOutRow + 4 * (lane >> 5) + pch;
  for (int i = lane; i < kMcPoolCh * (2 * kMcPoolW - 1); i += 64) {    // the border of the wave's eight planes, once
    const int ch = i / (2 * kMcPoolW - 1), r = i - ch * (2 * kMcPoolW - 1);
    pool[pool_base(ch) + (r < kMcPoolW ? r : (r - kMcPoolW + 1) * kMcPoolW)] = -INFINITY;
  }
  for (int i = tid; i < 2 * kMcMask; i += 256) lds[i] = 0.f;           // the padding of both mask buffers stays zero for good
  for (int i = tid; i < 128; i += 256) { par[128 + i] = scale[i]; par[256 + i] = shift[i]; }
  __syncthreads();

  auto mask_slot = [](int i) {                                         // element i of [2][27][27] -> its parity sub-plane
    const int ci = i / 729, r = i - ci * 729, y = r / 27 + 3, x = r - (y - 3) * 27 + 3;
    return ci * kMcPlane + ((y & 1) * 2 + (x & 1)) * kMcSub + (y >> 1) * kMcSubW + (x >> 1);
  };
  // pair p's masks: a batch of clips may leave them in per-clip tensors (mask_off, written by pair_prep_kernel)
  auto mask_base = [&](int q) { return masks + (mask_off ? mask_off[q] : (int64_t)q * 1458); };
  float pre[6];                                                        // the next pair's masks on their way to LDS
  auto load_masks = [&](int p) {
    const float* mp = mask_base(p < P ? p : 0);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = tid + 256 * q;
      pre[q] = (p < P && i < 1458) ? mp[i] : 0.f;
    }
  };
  auto store_masks = [&](float* mbuf) {
    // (opaque copy of the thread id: otherwise hipcc hoists the six padded offsets out of the pair loop and SPILLS them --
    // the reloads, each with a vmcnt(0) wait, cost more than recomputing six divisions by constants)
    int g_ = tid;
    asm volatile("" : "+v"(g_));
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = g_ + 256 * q;
      if (i < 1458) mbuf[mask_slot(i)] = pre[q];
    }
  };
  int p = blockIdx.x;
  load_masks(p);
  store_masks(lds);
  __syncthreads();

  f32x16 acc[7];
  // ---- MFMA phase of one pair ------------------------------------------------------------------------------------------
  auto conv = [&](const float* mbuf) {
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // K loop in units of two taps, software-pipelined by hand: the 14 reads of unit u + 1 are issued between the 14 MFMAs
    // of unit u (one read per MFMA; every address is the lane's base + an immediate), so a wave alone keeps the matrix pipe
    // fed -- hipcc's own schedule of the plain loop nest read, waited for lgkmcnt(0) and issued two or three MFMAs, 180
    // times per pair.
    int lb = lb0;                             // opaque: the address register is rebuilt HERE, per pair -- hoisted out of
    asm volatile("" : "+v"(lb));              // the pair loop hipcc materialises the 7 x 50 addresses and spills them
    const float* mb = mbuf + lb;
    float bq[2][7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      bq[0][j][0] = mb[2 * kMcSubW * j + tap_offset(0)];
      bq[0][j][1] = mb[2 * kMcSubW * j + tap_offset(1)];
    }
#pragma unroll
    for (int u = 0; u < kMcUnits; ++u) {
      if (u + 1 < kMcUnits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          bq[(u + 1) & 1][j][0] = mb[2 * kMcSubW * j + tap_offset(2 * u + 2)];
          // tap 49 is the bias tap: its weights are (bias[channel], 0) for the two lane halves, its operand is 1
          bq[(u + 1) & 1][j][1] = 2 * u + 3 < 49 ? mb[2 * kMcSubW * j + tap_offset(2 * u + 3)] : 1.f;
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 7; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u + tt) >> 2][(2 * u + tt) & 3], bq[u & 1][j][tt], acc[j], 0, 0, 0);
      if (u + 1 < kMcUnits) {
#pragma unroll
        for (int j = 0; j < 14; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- epilogue of one pair: bias -> ReLU -> BN, then 3x3/2 max pooling, 8 channels of this wave at a time ----------------
  // accumulator register 4q + r of column block j = channel 8q + r + 4*half, position 32j + fr
  auto epilogue = [&](int p) {
#ifdef STTRAN_MC_TRACE
    const bool fine_on = blockIdx.x == 0 && wave == 0 && lane == 0 && p == (int)gridDim.x;   // second pair of workgroup 0
#endif
    MC_FINE(0);
    float* dst = c2 + (int64_t)p * (49 * 128) + wave * 32;
    int pw = pw0, pb = pb0, ob = ob0;         // opaque, as in conv(): one base register each, the rest are immediates
    asm volatile("" : "+v"(pw), "+v"(pb), "+v"(ob));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cs[4], ct[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = wave * 32 + 8 * q + r + 4 * fh;
        cs[r] = par[128 + ch]; ct[r] = par[256 + ch];
      }
      MC_FINE(1 + 5 * q);
      if (fr < 28) {                          // (columns 28..31 of every block are padding)
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)         // channel r + 4 fh of the round: plane pool_base(r) (+ pool_base(4) fh, in pw)
            pool[pool_base(r) + pw + 2 * kMcPoolW * j] = relu_nan(acc[j][4 * q + r]) * cs[r] + ct[r];   // (bias: tap 49)
      }
      // `pool` is private to this wave and a wave's LDS operations execute in program order: the lanes' writes above are
      // visible to the reads below without a workgroup barrier; the compiler must keep the order
      MC_FINE(2 + 5 * q);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MC_FINE(3 + 5 * q);
#pragma unroll
      for (int i = 0; i < 7; ++i) {           // pooled row py = i of the round's eight channels
        const float* s = pool + pb + 2 * kMcPoolW * i;
        const float s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[kMcPoolW], s4 = s[kMcPoolW + 1], s5 = s[kMcPoolW + 2],
                    s6 = s[2 * kMcPoolW], s7 = s[2 * kMcPoolW + 1], s8 = s[2 * kMcPoolW + 2];
        // NaN-propagating maximum of the nine (torch's max_pool2d returns NaN if the window holds one): four v_max3 and
        // five unordered compares -- not eight compare / select chains
        // (inline asm: fmaxf() makes hipcc canonicalise every loaded operand with a v_max_f32 x, x first)
        float m0, m1, m2, m;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(s0), "v"(s1), "v"(s2));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(s3), "v"(s4), "v"(s5));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(s6), "v"(s7), "v"(s8));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m0), "v"(m1), "v"(m2));
        const bool un = __builtin_isunordered(s0, s1) | __builtin_isunordered(s2, s3) | __builtin_isunordered(s4, s5) |
                        __builtin_isunordered(s6, s7) | __builtin_isunordered(s8, s8);
        if (pact) outb[ob + 7 * kMcOutRow * i + kMcPoolCh * q] = un ? __builtin_nanf("") : m;
      }
      MC_FINE(4 + 5 * q);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next round's writes stay behind these reads
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MC_FINE(5 + 5 * q);
    }
    // the wave's [49][32] result: 392 pieces of 16 bytes, 8 per position = one 128-byte line of C2
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int o = lane + 64 * i, pos = o >> 3, c = o & 7;
      if (o < 49 * 8) *reinterpret_cast<f32x4*>(dst + pos * 128 + 4 * c) = *reinterpret_cast<const f32x4*>(outb + pos * kMcOutRow + 4 * c);
    }
    MC_FINE(21);
  };

  for (int it = 0; p < P; p += gridDim.x, ++it) {
    const float* cur = lds + (it & 1) * kMcMask;
    float* nxt = lds + ((it + 1) & 1) * kMcMask;
    const int pn = p + gridDim.x;
#ifdef STTRAN_MC_TRACE
    const bool tr = blockIdx.x < 4 && it < 32 && wave == 0 && lane == 0;
    if (tr) g_mc_trace[((blockIdx.x * 2) * 32 + it) * 3 + 0] = __builtin_amdgcn_s_memtime();
#endif
    load_masks(pn);                                             // lands during the MFMAs
    conv(cur);
#ifdef STTRAN_MC_TRACE
    if (tr) g_mc_trace[((blockIdx.x * 2) * 32 + it) * 3 + 1] = __builtin_amdgcn_s_memtime();
#endif
    if (pn < P) store_masks(nxt);                               // before the epilogue's stores: vmcnt counts in order
    epilogue(p);
#ifdef STTRAN_MC_TRACE
    if (tr) g_mc_trace[((blockIdx.x * 2) * 32 + it) * 3 + 2] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
  }
}

}  // namespace

// w0p: conv.0.weight [128][2][7][7] re-ordered to [128][13 groups][2 channels][4 taps]; tap 49 = (conv.0.bias, 0), taps 50, 51 zero
// (`bias` is not read by the kernel any more: the argument stays for the launch interface)
hipError_t launch_mask_conv1_pool(hipStream_t s, const float* masks, const int64_t* mask_off, const float* w0p,
                                  const float* bias, const float* scale, const float* shift, float* c2, int P) {
  if (P <= 0) return hipSuccess;
  static DeviceMarks marks;
  constexpr int lds_bytes = kMcLdsFloats * 4;
  hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(mask_conv1_pool_kernel), lds_bytes);
  if (e != hipSuccess) return e;
  const int grid = std::min(P, 2 * std::max(num_cus(), 1));            // two 4-wave workgroups per CU (79 KB of LDS each)
  hipLaunchKernelGGL(mask_conv1_pool_kernel, dim3(grid), dim3(256), lds_bytes, s, masks, mask_off, w0p, bias, scale, shift, c2, P);
  return hipGetLastError();
}

}  // namespace sttran
