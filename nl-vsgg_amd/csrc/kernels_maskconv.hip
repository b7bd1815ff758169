// kernels_maskconv.hip -- first half of the spatial-mask branch of the pair fusion in ONE kernel:
//   Conv2d(2, 128, kernel 7, stride 2, padding 3) -> ReLU -> BatchNorm2d(128, eval) -> MaxPool2d(3, 2, 1)
//   masks [P, 2, 27, 27] -> C2 [P, 7, 7, 128] (channel-last)              (lib/sttran.py:337-341)
// The 14x14x128 convolution output (100 KB per pair) never reaches HBM: a workgroup computes it for one
// pair in MFMA accumulators, applies bias / ReLU / BN, pools through a small LDS buffer and stores the
// 7x7x128 result (25 KB per pair), channel-last: the layout the implicit-GEMM 3x3 convolution gathers best.
//
// Per pair the convolution is a GEMM  [128 channels] x [K = 2 x 49 taps] x [196 positions]:
//   * one wave owns 32 output channels and all 196 positions (7 MFMA 32x32 column blocks, 224 columns);
//   * its weights stay in registers for the whole launch (13 groups of 4 k per lane-half = 52 VGPRs):
//     K is ordered (tap, channel-of-the-lane-half), i.e. in every v_mfma_f32_32x32x2_f32 the lower 32
//     lanes carry input channel 0 and the upper 32 lanes input channel 1 of the same tap;
//   * the B operand is never materialised (no im2col): the pair's two 27x27 masks sit zero-padded to 33x33
//     in LDS, and B[k][n] is a single ds_read_b32 at  lane_base(n, half) + tap_offset, where the tap offset
//     is a compile-time immediate -- no address arithmetic and no bounds tests in the loop.
// Compute-bound on the MFMA pipe (K is only 98): 364 MFMAs per wave per pair.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace sttran {
namespace {

constexpr int kMcW = 33;                     // 27 + 2 * 3 padding
constexpr int kMcPlane = kMcW * kMcW;        // 1089 floats per padded input channel
constexpr int kMcMask = 2 * kMcPlane;        // one pair
constexpr int kMcGroups = 13;                // 52 taps (49 real) in groups of 4
constexpr int kMcPoolCh = 8;                 // channels pooled per round and wave
constexpr int kMcPoolW = 15;                 // a pooled channel's 14 x 14 conv map with a -inf row above and column left of it
constexpr int kMcPoolPlane = kMcPoolW * kMcPoolW;   // 225 floats (odd: channels start on different LDS banks)
constexpr int kMcLdsFloats = 2 * kMcMask + 4 * kMcPoolCh * kMcPoolPlane + 3 * 128;
constexpr int kMcStagger = 0;                // s_sleep(127) rounds of the second workgroup of a CU before its first pair

// float offset of tap t inside a padded plane (taps 49..51 are padding: weight 0, any valid address)
__host__ __device__ constexpr int tap_offset(int t) { return t < 49 ? (t / 7) * kMcW + (t % 7) : 0; }

__global__ void __launch_bounds__(256, 2)
mask_conv1_pool_kernel(const float* __restrict__ masks, const int64_t* __restrict__ mask_off, const float* __restrict__ w0p,
                       const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
                       float* __restrict__ c2, int P, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  float* mbuf = lds;                                                   // [2][kMcMask], double-buffered
  float* pool = lds + 2 * kMcMask + wave * (kMcPoolCh * kMcPoolPlane); // wave-private [8][15][15], row 0 / column 0 = -inf
  float* par = lds + 2 * kMcMask + 4 * kMcPoolCh * kMcPoolPlane;       // bias | scale | shift, [3][128]

  // weights of this lane: channel 32*wave + fr, k = (group, half, e)  (w0p is [128][104] in that order)
  f32x4 a[kMcGroups];
#pragma unroll
  for (int g = 0; g < kMcGroups; ++g)
    a[g] = *reinterpret_cast<const f32x4*>(w0p + (wave * 32 + fr) * (kMcGroups * 8) + g * 8 + fh * 4);
  // B base offsets: column n = 32 j + fr is output position (n / 14, n % 14); its receptive field starts at
  // padded row 2*oy, column 2*ox of the lane-half's input channel
  int lb[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int n = 32 * j + fr;
    lb[j] = fh * kMcPlane + (n < 196 ? (n / 14) * 2 * kMcW + (n % 14) * 2 : 0);
  }
  // pooling map of this lane: output o = lane + 64 i of the round's 49 x 8 block, channel fastest (the result is
  // stored channel-last, [pair][7][7][128], so that the 3x3 convolution behind it gathers 4 channels per load)
  // The 3x3 / stride-2 / padding-1 window of output (py, px) covers conv rows 2py-1 .. 2py+1 and columns 2px-1 .. 2px+1;
  // with the -inf border row / column in front of the map every window is nine unconditional reads at pbase + {0,1,2} +
  // {0,15,30} (MaxPool2d pads with -inf, lib/sttran.py:341).
  int pbase[7], pdst[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int o = lane + 64 * i, oo = o < kMcPoolCh * 49 ? o : 0;
    const int pos = oo / kMcPoolCh, ch = oo - pos * kMcPoolCh, py = pos / 7, px = pos - py * 7;
    pbase[i] = ch * kMcPoolPlane + 2 * py * kMcPoolW + 2 * px;
    pdst[i] = pos * 128 + ch;
  }
  // where this lane's conv outputs go: column n = 32 j + fr of the 14 x 14 map -> (row + 1, column + 1) of the bordered plane
  // (= n + n / 14 + 16; the seven quotients n / 14 <= 13 ride in one register, four bits each)
  unsigned pwq = 0;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int n = 32 * j + fr;
    pwq |= (unsigned)((n < 196 ? n : 0) / 14) << (4 * j);
  }
  const int pw0 = fr + kMcPoolW + 1;
  for (int i = lane; i < kMcPoolCh * (2 * kMcPoolW - 1); i += 64) {    // the border of the wave's eight planes, once
    const int ch = i / (2 * kMcPoolW - 1), r = i - ch * (2 * kMcPoolW - 1);
    pool[ch * kMcPoolPlane + (r < kMcPoolW ? r : (r - kMcPoolW + 1) * kMcPoolW)] = -INFINITY;
  }

  for (int i = tid; i < 2 * kMcMask; i += 256) mbuf[i] = 0.f;          // the padding stays zero for good
  for (int i = tid; i < 128; i += 256) { par[i] = bias[i]; par[128 + i] = scale[i]; par[256 + i] = shift[i]; }
  __syncthreads();
  auto mask_slot = [](int i) {                                         // element i of [2][27][27] -> padded offset
    const int ci = i / 729, r = i - ci * 729, y = r / 27, x = r - y * 27;
    return ci * kMcPlane + (y + 3) * kMcW + (x + 3);
  };
  // pair p's masks: a batch of clips may leave them in per-clip tensors (mask_off, written by pair_prep_kernel)
  auto mask_base = [&](int q) { return masks + (mask_off ? mask_off[q] : (int64_t)q * 1458); };
  int p = blockIdx.x;
  if (p < P) {
    const float* mp = mask_base(p);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = tid + 256 * q;
      if (i < 1458) mbuf[mask_slot(i)] = mp[i];
    }
  }
  __syncthreads();

  // STAGGER.  Two workgroups share a CU (one wave of each per SIMD) and run the same program on equal work: started
  // together they stay in lockstep -- both in their MFMA phase (the matrix pipe shared), then both in their epilogue
  // (activation, pooling through LDS, stores: vector ALU and LDS only, the matrix pipe idle: it was busy 63 % of the
  // kernel, r5 PMC).  The workgroup that was placed SECOND on its CU (its LDS allocation does not start at 0) sleeps
  // for about one MFMA phase before its first pair; the offset then persists (neither workgroup gains on the other in
  // a period), so one's epilogue runs under the other's MFMAs.
  if (stagger > 0) {
    const unsigned lds_base = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 6);     // HW_REG_LDS_ALLOC.LDS_BASE
    if (lds_base != 0)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);                       // 127 x 64 clocks each
  }

  for (int it = 0; p < P; p += gridDim.x, ++it) {
    const float* cur = mbuf + (it & 1) * kMcMask;
    float* nxt = mbuf + ((it + 1) & 1) * kMcMask;
    const int pn = p + gridDim.x;
    float pre[6];
    const float* mpn = mask_base(pn < P ? pn : p);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = tid + 256 * q;
      pre[q] = (pn < P && i < 1458) ? mpn[i] : 0.f;
    }

    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // K loop in units of two taps (one ds_read2_b32 per column block and unit: both tap offsets are immediates), software-
    // pipelined by hand: the 7 reads of unit u + 1 are issued between the 14 MFMAs of unit u (one read per two MFMAs), so a
    // wave alone keeps the matrix pipe fed -- hipcc's own schedule of the plain loop nest read, waited for lgkmcnt(0) and
    // issued two or three MFMAs, 180 times per pair (r5: matrix pipe busy 63 % of the kernel).  Taps 50 and 51 (zero
    // weights: K is padded to groups of 4 for the register layout of `a`) are not executed: 25 units, 350 MFMAs.
    constexpr int kUnits = 25;
    float bq[2][7][2];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      bq[0][j][0] = cur[lb[j] + tap_offset(0)];
      bq[0][j][1] = cur[lb[j] + tap_offset(1)];
    }
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
      if (u + 1 < kUnits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          bq[(u + 1) & 1][j][0] = cur[lb[j] + tap_offset(2 * u + 2)];
          bq[(u + 1) & 1][j][1] = cur[lb[j] + tap_offset(2 * u + 3)];
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 7; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u + tt) >> 2][(2 * u + tt) & 3], bq[u & 1][j][tt], acc[j], 0, 0, 0);
      if (u + 1 < kUnits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // bias -> ReLU -> BN, then 3x3/2 max pooling, 8 channels of this wave at a time.
    // accumulator register 4q + r of column block j = channel 8q + r + 4*half, position 32j + fr
    float* dst = c2 + (int64_t)p * (49 * 128) + wave * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cb[4], cs[4], ct[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = wave * 32 + 8 * q + r + 4 * fh;
        cb[r] = par[ch]; cs[r] = par[128 + ch]; ct[r] = par[256 + ch];
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int n = 32 * j + fr;
        if (n < 196) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pool[(r + 4 * fh) * kMcPoolPlane + pw0 + 32 * j + (int)((pwq >> (4 * j)) & 15u)] =
                relu_nan(acc[j][4 * q + r] + cb[r]) * cs[r] + ct[r];
        }
      }
      // `pool` is private to this wave and a wave's LDS operations execute in program order: the lanes' writes above are
      // visible to the reads below without a workgroup barrier (rounds 1-2 had __syncthreads() here and behind the reads:
      // eight barriers per pair that only made the four waves wait for each other); the compiler must keep the order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int o = lane + 64 * i;
        if (o < kMcPoolCh * 49) {
          const float* s = pool + pbase[i];
          const float s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[kMcPoolW], s4 = s[kMcPoolW + 1], s5 = s[kMcPoolW + 2],
                      s6 = s[2 * kMcPoolW], s7 = s[2 * kMcPoolW + 1], s8 = s[2 * kMcPoolW + 2];
          // NaN-propagating maximum of the nine (torch's max_pool2d returns NaN if the window holds one): three v_max3 +
          // one, and five unordered compares -- not eight compare / select chains
          // (inline asm: fmaxf() makes hipcc canonicalise every loaded operand with a v_max_f32 x, x first)
          float m0, m1, m2, m;
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(s0), "v"(s1), "v"(s2));
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(s3), "v"(s4), "v"(s5));
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(s6), "v"(s7), "v"(s8));
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m0), "v"(m1), "v"(m2));
          const bool un = __builtin_isunordered(s0, s1) | __builtin_isunordered(s2, s3) | __builtin_isunordered(s4, s5) |
                          __builtin_isunordered(s6, s7) | __builtin_isunordered(s8, s8);
          dst[q * kMcPoolCh + pdst[i]] = un ? __builtin_nanf("") : m;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next round's writes stay behind these reads
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (pn < P) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int i = tid + 256 * q;
        if (i < 1458) nxt[mask_slot(i)] = pre[q];
      }
    }
    __syncthreads();
  }
}

}  // namespace

// w0p: conv.0.weight [128][2][7][7] re-ordered to [128][13 groups][2 channels][4 taps], taps 49..51 zero
hipError_t launch_mask_conv1_pool(hipStream_t s, const float* masks, const int64_t* mask_off, const float* w0p,
                                  const float* bias, const float* scale, const float* shift, float* c2, int P) {
  if (P <= 0) return hipSuccess;
  static DeviceMarks marks;
  constexpr int lds_bytes = kMcLdsFloats * 4;
  hipError_t e = marks.raise_lds(reinterpret_cast<const void*>(mask_conv1_pool_kernel), lds_bytes);
  if (e != hipSuccess) return e;
  const int grid = std::min(P, 2 * std::max(num_cus(), 1));
  static const int stagger = [] { const char* e = exp_env("STTRAN_MC_STAGGER"); return e ? atoi(e) : kMcStagger; }();
  hipLaunchKernelGGL(mask_conv1_pool_kernel, dim3(grid), dim3(256), lds_bytes, s, masks, mask_off, w0p, bias, scale, shift, c2, P,
                     P > grid ? stagger : 0);
  return hipGetLastError();
}

}  // namespace sttran
