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
//     in LDS, and B[k][n] is a single LDS read at  lane_base(n, half) + tap_offset, where the tap offset
//     is a compile-time immediate -- no address arithmetic and no bounds tests in the loop.
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

constexpr int kMcW = 33;                     // 27 + 2 * 3 padding
constexpr int kMcPlane = kMcW * kMcW;        // 1089 floats per padded input channel
constexpr int kMcMask = 2 * kMcPlane;        // one pair
constexpr int kMcGroups = 13;                // 52 taps (49 real) in groups of 4 (the register layout of the weights)
constexpr int kMcUnits = 25;                 // executed K: 25 units of two taps (tap 49 has zero weights; 50, 51 are skipped)
constexpr int kMcPoolCh = 8;                 // channels pooled per round and wave
constexpr int kMcPoolW = 15;                 // a pooled channel's 14 x 14 conv map with a -inf row above and column left of it
constexpr int kMcPoolPlane = kMcPoolW * kMcPoolW;   // 225 floats (odd: channels start on different LDS banks)
constexpr int kMcWaves = 4;
constexpr int kMcOutRow = 40;                // a wave's pooled result waits in LDS as [49 positions][32 channels], rows 40 floats apart
constexpr int kMcOut = 49 * kMcOutRow;       // (8 q + ch + 8 (pos % 4): the 32 lanes of a write hit 32 banks)
constexpr int kMcLdsFloats = 2 * kMcMask + kMcWaves * (kMcPoolCh * kMcPoolPlane + kMcOut) + 3 * 128;

// float offset of tap t < 49 inside a padded plane
__host__ __device__ constexpr int tap_offset(int t) { return (t / 7) * kMcW + (t % 7); }

__global__ void __launch_bounds__(256, 2)
mask_conv1_pool_kernel(const float* __restrict__ masks, const int64_t* __restrict__ mask_off, const float* __restrict__ w0p,
                       const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
                       float* __restrict__ c2, int P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: scalar registers
  const int fr = lane & 31, fh = lane >> 5;
  float* pool = lds + 2 * kMcMask + wave * (kMcPoolCh * kMcPoolPlane);  // wave-private [8][15][15], row 0 / column 0 = -inf
  float* outb = lds + 2 * kMcMask + kMcWaves * kMcPoolCh * kMcPoolPlane + wave * kMcOut;   // wave-private [49][40]
  float* par = lds + 2 * kMcMask + kMcWaves * (kMcPoolCh * kMcPoolPlane + kMcOut);            // bias | scale | shift, [3][128]

  // weights of this lane: channel 32*wave + fr, k = (group, half, e)  (w0p is [128][104] in that order)
  f32x4 a[kMcGroups];
#pragma unroll
  for (int g = 0; g < kMcGroups; ++g)
    a[g] = *reinterpret_cast<const f32x4*>(w0p + (wave * 32 + fr) * (kMcGroups * 8) + g * 8 + fh * 4);
  // Per-lane maps, kept as a few packed registers and expanded where they are used (the accumulators, the weights and the
  // masks in flight leave ~40 registers for everything else):
  //  * conv column n = 32 j + fr is output position (oy, ox) = (n / 14, n % 14).  The seven quotients oy <= 13 ride in `oyq`,
  //    four bits each; columns >= 196 (j = 6, fr >= 4) are padding and take oy = 0.
  //    B base of the column (its receptive field starts at padded row 2 oy, column 2 ox of the lane-half's channel):
  //        fh * 1089 + 66 oy + 2 ox = (fh * 1089 + 2 fr) + 64 j + 38 oy
  //    where its activation goes in the wave's pooling plane ((oy + 1, ox + 1) of the 15 x 15 bordered map):
  //        (fr + 16) + 32 j + oy
  //  * pooling: output o = lane + 64 i of a round's 49 x 8 block, channel fastest (the result is stored channel-last,
  //    [pair][7][7][128], so that the 3x3 convolution behind it gathers 4 channels per load): channel o % 8 = lane % 8,
  //    position pos = lane / 8 + 8 i = 7 py + px.  The 3x3 / stride-2 / padding-1 window of (py, px) covers conv rows
  //    2py-1 .. 2py+1 and columns 2px-1 .. 2px+1; with the -inf border row / column in front of the map every window is nine
  //    unconditional reads at base + {0,1,2} + {0,15,30} (MaxPool2d pads with -inf, lib/sttran.py:341), base =
  //        ch * 225 + 30 py + 2 px = (ch * 225 + 2 (lane / 8)) + 16 i + 16 py      (py <= 6: three bits each in `pyq`)
  //    The pooled value goes to the wave's LDS result block at (lane / 8) * 40 + lane % 8 + 320 i + 8 q; when the four rounds
  //    are done the block leaves as WHOLE 128-byte lines (a position's 32 channels of this wave), 16 bytes per lane: round
  //    1-4's dword stores wrote a line in four 32-byte pieces, one per round, and the first piece of every line stalled the
  //    wave for the line's allocation (s_memtime trace: the first round's stores took 16 k cycles, the others 1.7 k).
  unsigned oyq = 0, pyq = 0;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int n = 32 * j + fr;
    oyq |= (unsigned)((n < 196 ? n : 0) / 14) << (4 * j);
    pyq |= (unsigned)(min((lane >> 3) + 8 * j, 48) / 7) << (3 * j);
  }
  const int lb0 = fh * kMcPlane + 2 * fr, pw0 = fr + kMcPoolW + 1;
  const int pb0 = (lane & 7) * kMcPoolPlane + 2 * (lane >> 3), ob0 = (lane >> 3) * kMcOutRow + (lane & 7);
  for (int i = lane; i < kMcPoolCh * (2 * kMcPoolW - 1); i += 64) {    // the border of the wave's eight planes, once
    const int ch = i / (2 * kMcPoolW - 1), r = i - ch * (2 * kMcPoolW - 1);
    pool[ch * kMcPoolPlane + (r < kMcPoolW ? r : (r - kMcPoolW + 1) * kMcPoolW)] = -INFINITY;
  }
  for (int i = tid; i < 2 * kMcMask; i += 256) lds[i] = 0.f;           // the padding of both mask buffers stays zero for good
  for (int i = tid; i < 128; i += 256) { par[128 + i] = scale[i]; par[256 + i] = shift[i]; }
  __syncthreads();

  auto mask_slot = [](int i) {                                         // element i of [2][27][27] -> padded offset
    const int ci = i / 729, r = i - ci * 729, y = r / 27, x = r - y * 27;
    return ci * kMcPlane + (y + 3) * kMcW + (x + 3);
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
    // K loop in units of two taps (one ds_read2_b32 per column block and unit: both tap offsets are immediates), software-
    // pipelined by hand: the 7 reads of unit u + 1 are issued between the 14 MFMAs of unit u (one read per two MFMAs), so a
    // wave alone keeps the matrix pipe fed -- hipcc's own schedule of the plain loop nest read, waited for lgkmcnt(0) and
    // issued two or three MFMAs, 180 times per pair.
    unsigned oy_ = oyq;                       // opaque: expanded HERE, per pair -- hoisted out of the pair loop the seven
    asm volatile("" : "+v"(oy_));             // addresses are spilled (and reloaded behind vmcnt(0) waits)
    int lb[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) lb[j] = lb0 + 64 * j + 38 * (int)((oy_ >> (4 * j)) & 15u);
    float bq[2][7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      bq[0][j][0] = mbuf[lb[j] + tap_offset(0)];
      bq[0][j][1] = mbuf[lb[j] + tap_offset(1)];
    }
#pragma unroll
    for (int u = 0; u < kMcUnits; ++u) {
      if (u + 1 < kMcUnits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          bq[(u + 1) & 1][j][0] = mbuf[lb[j] + tap_offset(2 * u + 2)];
          // tap 49 is the bias tap: its weights are (bias[channel], 0) for the two lane halves, its operand is 1
          bq[(u + 1) & 1][j][1] = 2 * u + 3 < 49 ? mbuf[lb[j] + tap_offset(2 * u + 3)] : 1.f;
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 7; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u + tt) >> 2][(2 * u + tt) & 3], bq[u & 1][j][tt], acc[j], 0, 0, 0);
      if (u + 1 < kMcUnits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
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
    unsigned oy_ = oyq, py_ = pyq;            // opaque, as in conv()
    asm volatile("" : "+v"(oy_), "+v"(py_));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cs[4], ct[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = wave * 32 + 8 * q + r + 4 * fh;
        cs[r] = par[128 + ch]; ct[r] = par[256 + ch];
      }
      MC_FINE(1 + 5 * q);
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int n = 32 * j + fr;
        if (n < 196) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pool[(r + 4 * fh) * kMcPoolPlane + pw0 + 32 * j + (int)((oy_ >> (4 * j)) & 15u)] =
                relu_nan(acc[j][4 * q + r]) * cs[r] + ct[r];            // (the bias came in through tap 49)
        }
      }
      // `pool` is private to this wave and a wave's LDS operations execute in program order: the lanes' writes above are
      // visible to the reads below without a workgroup barrier; the compiler must keep the order
      MC_FINE(2 + 5 * q);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MC_FINE(3 + 5 * q);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int o = lane + 64 * i;
        if (o < kMcPoolCh * 49) {
          const float* s = pool + pb0 + 16 * i + 16 * (int)((py_ >> (3 * i)) & 7u);
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
          outb[ob0 + 8 * kMcOutRow * i + kMcPoolCh * q] = un ? __builtin_nanf("") : m;
        }
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
