/* sttran_hip_debug.h -- test hooks and experiment switches of libsttran_hip.so.  NOT part of the drop-in boundary
 * (include/sttran_hip.h): nothing here has a counterpart in the reference, a maintainer binding the library never needs
 * it, and it may change between rounds.  Used by tests/ (kernel-level parity, guard-page tests), tools/gemm_bench.py
 * and the opt-in bf16x3 experiment.
 */
#ifndef STTRAN_HIP_DEBUG_H
#define STTRAN_HIP_DEBUG_H

#include "sttran_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* GEMM engine of the nn.Linear layers and the two convolutions (no reference counterpart; SURVEY.md 7 names both: "fp32 MFMA ...
 * or 3 x bf16 split").  STTRAN_GEMM_FP32_MFMA (default): exact fp32 on v_mfma_f32_16x16x4_f32 / 32x32x2_f32.
 * STTRAN_GEMM_BF16X3 (opt-in second engine): fp32 EMULATED on the bf16 matrix pipe -- every operand split into three bf16
 * planes, six cross products per element pair, fp32 accumulation; measured error against fp64 no larger than the exact
 * engine's.  Since round 6 on v_mfma_f32_16x16x32_bf16 with both operands as pre-split fragment-major planes
 * (csrc/gemm_bf16x3_t16.h, gemm_bf16x3_t16c.h): weights are split once (at the next forward), activations by their producer
 * (LayerNorm, the linear1 / linear2 epilogues) or by one split pass; the launches those tiles do not serve (vr_fc, the grouped
 * subj | obj FC) run on round 2's kernel (csrc/gemm_bf16x3.h, activations split by the A loader).  Attention, the 7x7 mask
 * convolution and the small GEMMs (fewer than 512 rows, or N < 128) stay on the exact engine. */
/* STTRAN_GEMM_BF16X3_ALL: the emulated engine for EVERY contraction it can take (N >= 128), whatever the row count -- the
 * form the parity tests use so that small fixtures exercise it too (BF16X3 keeps launches under 512 rows on the exact engine,
 * where the 256-row emulation tile would mostly compute padding). */
enum { STTRAN_GEMM_FP32_MFMA = 0, STTRAN_GEMM_BF16X3 = 1, STTRAN_GEMM_BF16X3_ALL = 2 };
int sttran_set_gemm_engine(SttranHandle* h, int32_t engine);

/* Kernel-level test hooks: each runs ONE kernel class on caller-provided device buffers so the
 * parity tests can check kernels in isolation (tests/test_kernels_gpu.py). */
/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias) (+ residual); tile_cfg 0 = auto, split_k 0 = auto. */
int sttran_debug_gemm(const float* A, const int32_t* a_rowidx, const float* W, const float* bias,
                      const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                      int32_t relu, int32_t tile_cfg, int32_t split_k, void* stream);
/* The product's GEMM path (select-free, two-deep prefetch): A rows `lda` and W rows `ldw` floats apart, both readable
 * (finite) up to the next multiple of 32 columns past K, and W zero there -- how the library stores every nn.Linear
 * weight and lays out its workspace (csrc/gemm_f32_mfma.h, B_KMAJOR_PAD). */
int sttran_debug_gemm_padded(const float* A, int64_t lda, const int32_t* a_rowidx, const float* W, int64_t ldw,
                             const float* bias, const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                             int32_t relu, int32_t tile_cfg, void* stream);
/* EXPERIMENT (not on the product's default path): the same GEMM with fp32 EMULATED on the bf16 matrix pipe -- operands
 * split into three bf16 planes, six cross products per element pair on v_mfma_f32_32x32x16_bf16, fp32 accumulation
 * (csrc/gemm_bf16x3.h).  W [N,K] fp32 (row stride ldw) is split into planes inside every call; with
 * STTRAN_X3_CACHE_PLANES=1 in the environment the planes of the last W pointer are reused (timing runs). */
int sttran_debug_gemm_emulated(const float* A, int64_t lda, const int32_t* a_rowidx, const float* W, int64_t ldw,
                         const float* bias, const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                         int32_t relu, void* stream);
/* The same product through the second-generation bf16x3 kernel (csrc/gemm_bf16x3_t16.h: v_mfma_f32_16x16x32_bf16 on 128 x 176 /
 * 128 x 128 tiles, both operands pre-split into fragment-major planes).  N must be a multiple of 176 or 128. */
int sttran_debug_gemm_emulated_t16(const float* A, int64_t lda, const int32_t* a_rowidx, const float* W, int64_t ldw,
                         const float* bias, const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                         int32_t relu, void* stream);
/* tools/x3_bench.py: `iters` back-to-back launches of the 16x16x32 bf16x3 kernel on operands split once (us[0], microseconds
 * per launch) and of the activation split alone (us[1]).  Device pointers; lda, ldw multiples of 4. */
int sttran_debug_x3t16_bench(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* residual,
                             float* C, int64_t M, int64_t N, int64_t K, int32_t iters, double* us);
/* Calibration: fp32-MFMA rate (TFLOP/s) this device sustains on a register-only MFMA loop. */
int sttran_debug_mfma_peak(int32_t iters, double* tflops);
/* Test allocator: `bytes` of device memory (16-byte aligned) that END at the end of a mapping; the address range behind
 * it is reserved and unmapped, so an access past the buffer is a GPU memory fault, not a silent read of a neighbour.
 * `cookie` goes to sttran_debug_guarded_free.  STTRAN_ERR_HIP if the driver has no virtual-memory API. */
int sttran_debug_guarded_alloc(size_t bytes, void** ptr, void** cookie);
int sttran_debug_guarded_free(void* cookie);
/* on != 0: sttran_debug_guarded_free hands the address range back too (hipMemAddressFree).  Off by default: a range that is
 * reserved again and mapped to new memory can be reached through its old translation on this driver
 * (tools/experiments/vmm_reuse_probe.py is the only caller). */
int sttran_debug_guarded_return_addresses(int32_t on);
/* The tile id (1..8, see csrc/kernels.h) the planner picks for an [M,N,K] nn.Linear GEMM on the current device. */
int sttran_debug_plan_tile(int64_t M, int64_t N, int64_t K);
/* y[r,:] = LayerNorm(x[r,:]) * gamma + beta, eps 1e-5 (lib/transformer.py:15-16). */
int sttran_debug_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                           int64_t rows, int64_t dim, void* stream);
/* First half of the spatial-mask branch in one kernel (lib/sttran.py:337-341): Conv2d(2, 128, 7, stride 2, padding 3) ->
 * ReLU -> BatchNorm2d(128, eval) -> MaxPool2d(3, 2, 1).  masks [P,2,27,27] (or, mask_off != NULL: pair p's masks start at
 * masks + mask_off[p]); w0p = conv.0.weight packed [128][13 tap groups][2 channels][4 taps], tap 49 = (conv.0.bias, 0);
 * scale / shift = the folded eval-mode BatchNorm; c2 [P,7,7,128] channel-last.  All device pointers. */
int sttran_debug_mask_conv1_pool(const float* masks, const int64_t* mask_off, const float* w0p, const float* scale,
                                 const float* shift, float* c2, int32_t P, void* stream);
/* Multi-head attention core on packed qkv [tokens, 3*dim] over sequences given by
 * (seq_off, seq_len) device arrays; out [tokens, dim].  nn.MultiheadAttention semantics
 * (q scaled by 1/sqrt(dim/nhead), softmax over the keys of the same sequence). */
int sttran_debug_attention(const float* qkv, const int32_t* seq_off, const int32_t* seq_len,
                           int32_t num_seq, int32_t max_len, float* out, int64_t tokens,
                           int32_t dim, int32_t nhead, void* stream);

/* DSG-DETR class sequences as the forward builds them on the device (lib/dsg_detr.py:545-555; csrc/kernels_front.hip
 * dsg_layout_kernel): clip_start [num_clips + 1] = pair range of every clip; outputs dec_off / dec_len
 * [num_clips * num_classes] (slot = clip * num_classes + class), dec_src (token -> pair), need (token -> position index,
 * handed out by position like the reference), out_src (pair -> num_pairs + token), each [num_pairs]; scratch4p
 * [4 * num_pairs] ints; err_flag: bit 0 = index out of range, bit 1 = position index >= pe_rows.  All device pointers. */
int sttran_debug_dsg_layout(const int64_t* pair_idx, const int64_t* labels, int64_t num_boxes, const int32_t* clip_start,
                            int32_t num_clips, int32_t num_classes, int64_t num_pairs, int32_t pe_rows, int32_t* dec_off,
                            int32_t* dec_len, int32_t* dec_src, int32_t* need, int32_t* out_src, int32_t* scratch4p,
                            int32_t* err_flag, void* stream);
/* sttran_debug_attention for sequences whose lengths only the device knows: len_bound >= every seq_len[i]; one launch per
 * length class ((0,32], (32,80], (80,len_bound]), empty slots allowed. */
int sttran_debug_attention_classes(const float* qkv, const int32_t* seq_off, const int32_t* seq_len, int32_t num_seq,
                                   int32_t len_bound, float* out, int32_t dim, int32_t nhead, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STTRAN_HIP_DEBUG_H */
