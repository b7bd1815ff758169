// kernels.h -- host-side launch interface of the HIP kernels (internal to libsttran_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32_mfma.h"

namespace sttran {

// ---- per-device launch state -----------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count belong to a DEVICE, and sttran_create() accepts
// any ordinal: every launcher keeps its "attribute already raised to N bytes" mark and the planner its CU count
// per device ordinal (relaxed atomics: two threads racing on the same device set the same value twice).
constexpr int kMaxDevices = 64;
inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}
struct DeviceMarks {
  int v[kMaxDevices] = {};
  // raise `fn`'s dynamic-LDS limit on the current device to `bytes` unless a previous call already did
  hipError_t raise_lds(const void* fn, int bytes) {
    const int d = current_device();
    if (__atomic_load_n(&v[d], __ATOMIC_RELAXED) >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_store_n(&v[d], bytes, __ATOMIC_RELAXED);
    return e;
  }
};
int num_cus();   // compute units of the current device (cached per ordinal)

// Tuning / A-B knobs come from the environment in EXPERIMENT builds only (make EXTRA=-DSTTRAN_GEMM_EXPERIMENT: what
// tools/gemm_bench.py and tools/experiments/ use).  The product library reads NO environment variable on the forward path
// (the one exception, STTRAN_GUARD_WORKSPACE, switches the test allocator of tests/test_guarded_buffers_gpu.py and is
// read once, at the first allocation).
inline const char* exp_env(const char* name) {
#ifdef STTRAN_GEMM_EXPERIMENT
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// ---- GEMM ------------------------------------------------------------------------------
// tiles of gemm_f32_mfma.h (ids are part of the sttran_debug_gemm test hook: keep them stable)
// 5: gemm_f32_t16.h (16x16x4 MFMA blocks; N % 176 == 0, padded operands, vector epilogue only)
// 6 was a 256 x 176 form of the 16x16x4 family (never faster than 128 x 176, 66 spilled VGPRs): removed in round 4, id retired
enum { TILE_AUTO = 0, TILE_256x128 = 1, TILE_128x128 = 2, TILE_64x64 = 3, TILE_128x64 = 4, TILE_128x176 = 5, TILE_RETIRED_6 = 6,
       TILE_T128x128 = 7, TILE_COUNT = 8 };
struct GemmPlan { int tile; int splitk; };
GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, int force_tile, int force_split);
size_t gemm_slab_floats(const GemmPlan& p, int64_t M, int64_t N);
size_t gemm_slab_floats_max();   // park space of the stream-K schedule, any tile
size_t gemm_slab_bytes();        // what a `slab` argument must point to

// padded: both operands can be read up to ceil32(K) columns per row and B is zero there (gemm_f32_mfma.h, B_KMAJOR_PAD)
hipError_t gemm_linear(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                       const EpiLinear& epi, GemmPlan plan, float* slab, int padded);
// the tile gemm_linear really launches for these operands (plan.tile, or TILE_256x128 when a 16x16x4 tile's contract is not met)
int gemm_effective_tile(const GemmOperand& A, const GemmOperand& B, int N, int K, const EpiLinear& epi, GemmPlan plan, int padded);
hipError_t gemm_heads(hipStream_t s, const GemmOperand& A, const GemmOperand& B, int M, int N, int K,
                      const EpiHeads& epi, GemmPlan plan, float* slab);
hipError_t launch_mask_conv2(hipStream_t s, const float* w4, const float* c2, const EpiConvRelBn& epi, int P,
                             float* slab);

// the same two convolutions on the 16x16x4 kernel structure (gemm_f32_t16c.h): the product path since round 3
// (tile_base: the launch covers the 128-column tiles from there on -- the ones behind the fused launch's)
hipError_t launch_union_conv_t16(hipStream_t s, const float* U, const int64_t* u_off, const float* W, const float* bias, float* V,
                                 int P, int K, float* slab, int tile_base = 0);
hipError_t launch_mask_conv2_t16(hipStream_t s, const float* w4, const float* c2, const float* bias, const float* scale,
                                 const float* shift, float* V, int P, float* slab, int tile_base = 0);
// both of them in one pass over a tile (pair_conv_fused_kernel, gemm_f32_t16c.h): the first pair_convs_fused_tiles(P) column
// tiles of a launch (0 = the launch is too small for it); the rest goes through the two launches above with tile_base
int pair_convs_fused_tiles(int P);
hipError_t launch_pair_convs_fused_t16(hipStream_t s, const float* w4, const float* c2, const float* bias4, const float* scale,
                                       const float* shift, const float* U, const int64_t* u_off, const float* W, const float* bias1,
                                       float* V, int P, int K, int ntiles);

hipError_t launch_mfma_peak(hipStream_t s, float* out, int iters, int blocks);

// bf16x3 fp32 emulation (gemm_bf16x3.h; experiment): weight planes [3][rows][ldp] bf16 made by split_planes
hipError_t split_planes(hipStream_t s, const float* W, int64_t ld, int rows, int cols, void* planes, int64_t ldp);
// planes points at the first needed weight row of plane 0; plane_stride = elements between planes (rows_total * ldp)
hipError_t gemm_linear_x3(hipStream_t s, const GemmOperand& A, const void* planes, int64_t ldp, int64_t plane_stride, int M,
                          int N, int K, const EpiLinear& epi, float* slab);
// union_func1 on the same engine: planes = [3][256][K] bf16 of union_func1.weight
hipError_t launch_mask_conv2_x3(hipStream_t s, const void* planes, const float* c2, const float* bias, const float* scale,
                                const float* shift, float* V, int P, float* slab);
hipError_t launch_union_conv_x3(hipStream_t s, const float* U, const int64_t* u_off, const void* planes, const float* bias,
                                float* V, int P, int K, float* slab);

// second generation of the same engine (gemm_bf16x3_t16.h): both operands as FRAGMENT-MAJOR bf16 planes
// [row block 16][K block 32][plane 3][512] -- fm_planes_bytes(rows, K) bytes; split_fm makes them from fp32 rows (optionally
// gathered: rowoff = 64-bit element offsets, else rowidx), zero beyond M rows / K columns
size_t fm_planes_bytes(int64_t rows, int64_t K);
// weight = 1: the lane-major order of a weight (it passes through LDS); 0: an activation operand (row-major blocks)
hipError_t split_fm(hipStream_t s, const float* src, int64_t ld, const int32_t* rowidx, const int64_t* rowoff, int M, int K,
                    void* planes, int weight = 0);
int x3t16_tile(int N, const EpiLinear& epi);          // TILE_128x176 / TILE_T128x128, or 0 = shape / epilogue not served
// b_planes = the weight's planes at its first needed row block; b_row_blocks = row blocks available from there
hipError_t gemm_linear_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                             const EpiLinear& epi, float* slab);

// the same GEMM with out = act(A W^T + bias) written as the NEXT launch's fragment-major activation planes [M, N] (N % 32 == 0)
hipError_t gemm_act_planes_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                                 const float* bias, int relu, void* out_planes, float* slab);

// C = A W^T + bias + residual as fp32 rows AND as fragment-major planes [M, N] (N's tail to the next multiple of 32 is NOT
// written: the caller's buffer must hold zeros there -- hplanes does, LayerNorm writes that tail on every pass)
hipError_t gemm_res_planes_x3t16(hipStream_t s, const void* a_planes, const void* b_planes, int b_row_blocks, int M, int N, int K,
                                 const EpiLinear& epi, void* out_planes, float* slab);
// the two convolutions on the same engine (gemm_bf16x3_t16c.h): planes_fm = fragment-major planes (split_fm, weight = 1) of the
// [256, K] / [256, 1152] weight
hipError_t launch_union_conv_x3t16(hipStream_t s, const float* U, const int64_t* u_off, const void* planes_fm, const float* bias,
                                   float* V, int P, int K, float* slab, int tile_base = 0);
hipError_t launch_mask_conv2_x3t16(hipStream_t s, const void* planes_fm, const float* c2, const float* bias, const float* scale,
                                   const float* shift, float* V, int P, float* slab, int tile_base = 0);
// ... and both in one pass over a tile (pair_conv_fused_x3_kernel), as launch_pair_convs_fused_t16 above (tiles of 128 rows x 128
// channels here: tile_base counts those)
int pair_convs_fused_tiles_x3(int P);
hipError_t launch_pair_convs_fused_x3t16(hipStream_t s, const void* w4_planes_fm, const float* c2, const float* bias4, const float* scale,
                                         const float* shift, const float* U, const int64_t* u_off, const void* wu_planes_fm,
                                         const float* bias1, float* V, int P, int K, int ntiles);

// ---- fusion front-end (lib/sttran.py:381-399) ----------------------------------------------
// Where the inputs of one call live: n chunks, chunk c = the tensors of one clip as the caller passed them (SttranInputs'
// per-clip pointer tables) or ONE chunk = the whole contiguous batch.  All members are DEVICE arrays: the two prefix
// arrays have n + 1 entries, the pointer arrays n (boxes / dist: sgdet only).  pair_idx rows of a chunk are local to
// the chunk's boxes.
struct ChunkTable {
  int n;
  int base;                     // first chunk with pairs: per-pair offsets are relative to ITS features / union_feat / masks
  const int64_t* pair_start;
  const int64_t* box_start;
  const void* const* features;
  const void* const* pair_idx;
  const void* const* labels;
  const void* const* union_feat;
  const void* const* masks;
  const void* const* boxes;
  const void* const* dist;
};
// per pair: element offsets of its feature rows [2][P] / union_feat block [P] / spatial_masks block [P] from chunk 0's
// tensors, the two class-embedding column blocks of x, and (optional, DSG-DETR) object class + global subject row
hipError_t launch_pair_prep(hipStream_t s, const ChunkTable& tab, int P, int feat_dim, int num_classes, const float* emb1,
                            const float* emb2, int emb_dim, int64_t* feat_off, int64_t* union_off, int64_t* mask_off,
                            int* cls_of_pair, int* subj_of_pair, float* x, int ldx, int col_off, int* err_flag);
// Conv2d(2,128,k7,s2,p3) -> ReLU -> BN -> MaxPool(3,2,1) of the spatial masks in one kernel (lib/sttran.py:337-341):
// masks [P,2,27,27] -> c2 [P,128,7,7].  w0p = conv.0.weight re-ordered to [128][13][2][4] (tap group, input
// channel, tap in group; taps 49..51 zero).
// mask_off (optional): pair p's [2,27,27] block starts at masks + mask_off[p] floats (else at masks + 1458 p)
hipError_t launch_mask_conv1_pool(hipStream_t s, const float* masks, const int64_t* mask_off, const float* w0p,
                                  const float* bias, const float* scale, const float* shift, float* c2, int P);
// union_func1: V[p][c][hw] += W[c][:] . U[p][:][hw] + b[c]   (V already holds the mask-conv branch)
// u_off (optional): pair p's [K,7,7] block starts at U + u_off[p] floats (else at U + 49 K p)
hipError_t launch_union_conv(hipStream_t s, const float* U, const int64_t* u_off, const float* W, const float* bias, float* V,
                             int P, int K, float* slab);

// DSG-DETR class sequences built on the device (lib/dsg_detr.py:545-555): clip_start [num_clips + 1] pair ranges;
// dec_off / dec_len [num_clips * NC], dec_src / need / out_src [P], scratch4p [4 P] ints; err_flag bits 0 / 1;
// max_clip_pairs (0 = unknown) lets the kernel keep its per-token tables in LDS
hipError_t launch_dsg_layout(hipStream_t s, const int64_t* pair_idx, const int64_t* labels, int B, const int* clip_start,
                             int num_clips, int NC, int P, int pe_rows, int max_len, int* dec_off, int* dec_len, int* dec_src,
                             int* need, int* out_src, int* scratch4p, int* err_flag, int max_clip_pairs);

// Recall@K matching of one clip against its packed ground truth (lib/evaluation_recall.py:397-465,630-773):
// flags[g][metric*3 + k] = ground-truth relation g is hit within the first {10,20,50} predictions of
// metric {with constraint, no constraint, semi constraint}.  status: bit 0 = a frame has too many pairs,
// bit 1 = pair_idx out of range.
hipError_t launch_eval_recall(hipStream_t s, const float* att, const float* spa, const float* con,
                              const int64_t* pair_idx, const void* im_idx, int im_idx_i64, const float* boxes,
                              const int64_t* classes, const float* obj_scores, int P, int B, int na, int ns, int nc,
                              int F, const int32_t* gt_box_off, const float* gt_boxes, const int32_t* gt_classes,
                              const int32_t* gt_rel_off, const int32_t* gt_rels, double iou_thr, uint8_t* flags,
                              int32_t* status);
int eval_max_pairs_per_frame(int ncol);

// union boxes + soft box masks of each pair (lib/object_detector.py:110-124)
hipError_t launch_union_boxes_masks(hipStream_t s, const float* boxes, const int64_t* pair_idx, const float* im_idx,
                                    int P, int pool, float* union_boxes, float* masks);

// ---- transformer pieces ----------------------------------------------------------------------
// x rows ldx floats apart, y rows ldy floats apart (the workspace keeps [*, 1936] activations at a row stride of 1952)
// planes (optional, bf16x3 engine): y additionally leaves as fragment-major bf16 planes [rows, dim] (gemm_bf16x3_t16.h),
// columns dim .. ceil32(dim) zero
hipError_t launch_layernorm(hipStream_t s, const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                            int64_t ldy, int64_t rows, int dim, void* planes = nullptr);
// q_begin (optional, per sequence): compute only query rows [q_begin, len)
// qkv rows are 3*dim floats apart, out rows ldo floats apart
hipError_t launch_attention(hipStream_t s, const float* qkv, const int* seq_off, const int* seq_len,
                            const int* q_begin, int num_seq, int max_len, float* out, int64_t ldo, int dim, int nhead);
// the lengths are only known on the device (len_bound >= every one of them): one launch per length class, no read-back
hipError_t launch_attention_classes(hipStream_t s, const float* qkv, const int* seq_off, const int* seq_len, int num_seq,
                                    int len_bound, float* out, int64_t ldo, int dim, int nhead);
constexpr int kAttnChunkKeys = 480;  // keys per pass of the general attention kernel's LDS score block (longer sequences:
                                     // several passes with a running softmax -- no limit on the sequence length)
hipError_t launch_gather_rows(hipStream_t s, const float* src, int64_t lds, const int* idx, float* dst, int64_t ldd,
                              int64_t rows, int dim);

hipError_t launch_gather_add_rows(hipStream_t s, const float* src, int64_t lds, const int* idx, const float* table,
                                  int64_t ldt, const int* tidx, float* dst, int64_t ldd, int64_t rows, int dim);

// ---- ObjectClassifier sgdet+wks (lib/sttran.py:173-184) ---------------------------------------
// z[b] = [features[b] | distribution[b] @ E0 | ReLU(Linear(BN(center_size(box))))]   -> [B, 2376]
hipError_t launch_objcls_prep(hipStream_t s, const ChunkTable& tab, const float* E0, const float* pos_scale,
                              const float* pos_shift, const float* pos_w, const float* pos_b, float* z, int64_t ldz, int B,
                              int feat_dim, int ncls, int emb_dim);

// ---- SGDet without weak supervision (lib/sttran.py:185-283): kernels_objcls.hip ------------------------
size_t objcls_scratch_bytes(int64_t B, int T);
// clean_class + per-(frame, class) NMS + labels / scores / human / pairs.  Outputs need 4 * B rows (pairs: 4 * B too).
// Synchronises `s` once to return {rows, pairs, status} in host_out.
hipError_t launch_objcls_select(hipStream_t s, const float* boxes, const float* dist, const float* feats,
                                const int64_t* labels, int64_t B, int T, int ncol, int F, float thr, int ge, int64_t capacity,
                                float* o_boxes, float* o_dist, float* o_feats, float* o_score, int64_t* o_label, int* o_src,
                                int64_t* o_pair, float* o_im, int64_t* o_human, void* scratch, int32_t host_out[3]);
hipError_t launch_roi_align(hipStream_t s, const float* fmaps, int T, int C, int H, int W, const float* rois, int64_t P,
                            int pooled, float spatial_scale, int sampling_ratio, float* out);

}  // namespace sttran
