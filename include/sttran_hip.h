/* sttran_hip.h -- C ABI of the MI355X-native STTran relation-transformer hot path.
 *
 * The reference (rlqja1107/NL-VSGG) has no FFI boundary on this path: the path is the Python
 * call `pred = model(entry)` (tools/test_STTran.py:84) into `STTran.forward`
 * (lib/sttran.py:375-411) -> `transformer.forward` (lib/transformer.py:130-187).  This header is
 * the boundary a drop-in replacement introduces *under* that call; every entry point cites the
 * reference code it stands in for.  Plain pointers and sizes only -- no torch types.
 *
 * Threading: one handle per process/GPU; calls on one handle must be serialised by the caller (on the host; a forward on
 * another stream than the handle's previous one is ordered behind it on the device by the library).
 * Kernel-level test hooks and the experimental GEMM-engine switch are NOT part of this boundary: sttran_hip_debug.h.
 * sttran_forward only enqueues work on the supplied stream (no host synchronisation) when the
 * caller passes `frame_counts`; otherwise it reads `im_idx` back once (the reference itself
 * synchronises twice per frame, lib/transformer.py:138-140).
 * Errors: every function returns STTRAN_OK or a negative-free error code; no exceptions cross
 * the ABI; sttran_last_error() gives the message of the last failure on that handle.
 */
#ifndef STTRAN_HIP_H
#define STTRAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SttranHandle SttranHandle;

enum {
  STTRAN_OK = 0,
  STTRAN_ERR_INVALID = 1,  /* bad argument / struct_size / shape                          */
  STTRAN_ERR_HIP = 2,      /* a HIP runtime call failed                                   */
  STTRAN_ERR_EMPTY = 3,    /* no pairs: the reference passes `{}` on and crashes later
                              (tools/test_STTran.py:83-87); here it is an explicit error  */
  STTRAN_ERR_WEIGHTS = 4,  /* forward before all tensors of the state-dict were loaded    */
  STTRAN_ERR_ORDER = 5,    /* im_idx not sorted ascending (lib/transformer.py:138 assumes) */
  STTRAN_ERR_LIMIT = 6,    /* more pairs than 32-bit indexing allows, or (DSG-DETR) a class
                              sequence over more frames than the positional table has rows */
  STTRAN_ERR_INDEX = 7     /* sttran_sync_check: a kernel met a pair_idx / labels entry out of range (torch raises an
                              IndexError at lib/sttran.py:381-393; the kernels clamp the value and flag it) */
};

enum { STTRAN_MODE_PREDCLS = 0, STTRAN_MODE_SGCLS = 1, STTRAN_MODE_SGDET = 2 };
enum { STTRAN_DTYPE_F32 = 0, STTRAN_DTYPE_I64 = 1, STTRAN_DTYPE_I32 = 2 };
/* STTRAN: lib/sttran.py::STTran.  DSG_DETR: lib/dsg_detr.py::STTran (same fusion front-end and heads;
 * stock encoder layers per frame and per object-class sequence; sgdet mode only). */
enum { STTRAN_MODEL_STTRAN = 0, STTRAN_MODEL_DSG_DETR = 1 };

/* Constructor arguments of `STTran.__init__` (lib/sttran.py:316-318) that shape the compute. */
typedef struct SttranConfig {
  uint32_t struct_size;        /* = sizeof(SttranConfig) */
  int32_t device;              /* HIP device ordinal */
  int32_t mode;                /* STTRAN_MODE_*; sgdet means sgdet + is_wks (lib/sttran.py:173-184) */
  int32_t enc_layers;          /* enc_layer_num (1) */
  int32_t dec_layers;          /* dec_layer_num (3) */
  int32_t attention_classes;   /* 3  */
  int32_t spatial_classes;     /* 6  */
  int32_t contact_classes;     /* 17 */
  int32_t num_obj_classes;     /* len(obj_classes) = 37 incl. background */
  int32_t feat_dim;            /* 2048 */
  int32_t embed_dim;           /* 1936 (lib/sttran.py:358) */
  int32_t nhead;               /* 8 */
  int32_t ffn_dim;             /* 2048 */
  int32_t model;               /* STTRAN_MODEL_* */
} SttranConfig;

/* The `entry` dict read by STTran.forward.  All tensor pointers are DEVICE
 * pointers to contiguous row-major fp32 / int64 data; counts are host values.
 * A call may carry several clips back to back (num_clips > 1): frames are then numbered
 * consecutively over the whole batch and temporal windows never span a clip boundary.  With
 * num_clips == 1 this is exactly one reference call.
 *
 * Two ways to hand over a batch of clips:
 *  (1) contiguous: the seven tensor pointers below, every clip's rows concatenated (pair_idx holding batch-global box
 *      rows) -- what `pack_clips(entries)` builds by copying;
 *  (2) per-clip pointer tables (clip_union_feat != NULL): HOST arrays of num_clips DEVICE pointers, each clip's tensors
 *      exactly as its producer left them -- the reference's producer hands over one clip's `entry` at a time
 *      (tools/test_STTran.py:81-84, dataloader/wk_action_genome.py:622-627), so a batch of them is a list of separate
 *      allocations; nothing is copied or concatenated, the kernels resolve pair -> clip through a prefix table.  In this
 *      form every clip's pair_idx holds box rows LOCAL to that clip (its own, unmodified tensor), num_boxes / num_pairs are
 *      the totals, clip_num_boxes / clip_num_pairs the per-clip sizes, frame_counts is required (im_idx is not read), and
 *      the seven contiguous pointers are ignored.  Outputs are contiguous over the batch either way (clip c's rows start at
 *      the sum of the pair counts of the clips before it).  `pack_clips(entries, copy=False)` builds this form. */
typedef struct SttranInputs {
  uint32_t struct_size;          /* = sizeof(SttranInputs), or STTRAN_INPUTS_V1_SIZE (the form without the tables) */
  int32_t num_clips;             /* >= 1 */
  int64_t num_boxes;             /* B */
  int64_t num_pairs;             /* P */
  int32_t num_frames;            /* T over all clips; 0 = derive as im_idx[-1]+1 (D2H sync) */
  int32_t im_idx_dtype;          /* STTRAN_DTYPE_F32 (predcls) or STTRAN_DTYPE_I64 (sgdet) */
  const int32_t* clip_num_frames;/* HOST [num_clips]; may be NULL when num_clips == 1 */
  const int32_t* frame_counts;   /* HOST [num_frames] pairs per frame, or NULL (derive: sync) */
  const float* features;         /* [B, feat_dim]            entry['features']       */
  const int64_t* pair_idx;       /* [P, 2] global box rows   entry['pair_idx']       */
  const int64_t* labels;         /* [B]                      entry['labels']         */
  const float* union_feat;       /* [P, feat_dim, 7, 7] NCHW entry['union_feat']     */
  const float* spatial_masks;    /* [P, 2, 27, 27]           entry['spatial_masks']  */
  const void* im_idx;            /* [P] frame id of each pair, sorted ascending      */
  const float* boxes;            /* [B, 5] sgdet only        entry['boxes']          */
  const float* distribution;     /* [B, num_obj_classes-1] sgdet only                */
  /* ---- form (2): per-clip pointer tables, HOST arrays of num_clips device pointers (all NULL in form 1) ---- */
  const float* const* clip_features;        /* -> [B_c, feat_dim]           (16-byte aligned) */
  const int64_t* const* clip_pair_idx;      /* -> [P_c, 2] box rows local to clip c            */
  const int64_t* const* clip_labels;        /* -> [B_c]                                        */
  const float* const* clip_union_feat;      /* -> [P_c, feat_dim, 7, 7]                        */
  const float* const* clip_spatial_masks;   /* -> [P_c, 2, 27, 27]                             */
  const float* const* clip_boxes;           /* -> [B_c, 5]                  sgdet only         */
  const float* const* clip_distribution;    /* -> [B_c, num_obj_classes-1]  sgdet only         */
  const int64_t* clip_num_boxes;            /* HOST [num_clips] B_c, summing to num_boxes      */
  const int64_t* clip_num_pairs;            /* HOST [num_clips] P_c, summing to num_pairs; a clip's P_c must equal the
                                               sum of its frames' frame_counts                 */
} SttranInputs;
#define STTRAN_INPUTS_V1_SIZE 112u  /* offsetof(SttranInputs, clip_features): callers built against the round-2 header */

/* Tensors STTran.forward writes into `entry` (lib/sttran.py:404-409, :182).  Caller-allocated
 * device buffers.  The three *_tap pointers are optional (NULL) stage outputs for parity tests. */
typedef struct SttranOutputs {
  uint32_t struct_size;            /* = sizeof(SttranOutputs) */
  uint32_t reserved;
  float* attention_distribution;   /* [P, attention_classes]  raw logits          */
  float* spatial_distribution;     /* [P, spatial_classes]    sigmoid             */
  float* contacting_distribution;  /* [P, contact_classes]    sigmoid             */
  float* distribution;             /* [B, num_obj_classes] sgdet only, else NULL  */
  float* rel_features_tap;         /* [P, embed_dim]  lib/sttran.py:399           */
  float* local_output_tap;         /* [P, embed_dim]  lib/transformer.py:145      */
  float* global_output_tap;        /* [P, embed_dim]  lib/transformer.py:187      */
} SttranOutputs;

/* Per-kernel-class device time, measured with HIP events on the forward's own stream while
 * profiling is enabled (bench.py's roofline leg). */
#define STTRAN_PROF_CLASSES 8
enum {
  STTRAN_PROF_GEMM = 0,       /* gemm_f32_mfma (all Linear layers, conv3x3 as implicit GEMM) */
  STTRAN_PROF_UNION_CONV = 1, /* union_func1 1x1 conv                                      */
  STTRAN_PROF_MASK_CONV = 2,  /* conv7x7/2 -> ReLU -> BN -> max-pool of the spatial masks   */
  STTRAN_PROF_ATTENTION = 3,
  STTRAN_PROF_LAYERNORM = 4,
  STTRAN_PROF_INDEX = 5,      /* gather / scatter / embedding / split-K reduce             */
  STTRAN_PROF_OTHER = 6
};
typedef struct SttranProfile {
  uint32_t struct_size;
  uint32_t forwards;                       /* forwards accumulated since profile_reset     */
  double ms[STTRAN_PROF_CLASSES];          /* summed launch durations per class            */
  double flops[STTRAN_PROF_CLASSES];       /* algorithmic FLOPs (2*M*N*K, unpadded)        */
  double bytes[STTRAN_PROF_CLASSES];       /* algorithmic bytes (operands + outputs once)  */
  uint64_t launches[STTRAN_PROF_CLASSES];
} SttranProfile;

/* `STTran(...)` constructor, lib/sttran.py:316-372 (weights are NOT initialised here). */
int sttran_create(const SttranConfig* cfg, SttranHandle** out);
/* `model.load_state_dict(ckpt['state_dict'], strict=False)`, tools/test_STTran.py:51-52.
 * `key` is the reference state-dict key; data is fp32 (or int64 for
 * num_batches_tracked, ignored).  `on_device` != 0 means `data` is a device pointer.
 * Unknown keys are ignored (strict=False) and reported through the return of
 * sttran_missing_keys. */
int sttran_load_tensor(SttranHandle* h, const char* key, const void* data, const int64_t* shape,
                       int32_t ndim, int32_t dtype, int32_t on_device);
/* Derive fused parameters (BatchNorm scale/shift, position-embedding biases, packed heads);
 * called implicitly by the first forward.  Returns STTRAN_ERR_WEIGHTS if keys are missing. */
int sttran_finalize_weights(SttranHandle* h);
/* Writes a '\n'-separated list of still-missing state-dict keys into buf; returns their count. */
int sttran_missing_keys(SttranHandle* h, char* buf, int64_t buflen);
/* Pre-size the workspace (otherwise grown on demand by forward; growth synchronises). */
int sttran_reserve(SttranHandle* h, int64_t max_pairs, int64_t max_boxes);
/* `pred = model(entry)` under torch.no_grad(), tools/test_STTran.py:84 ->
 * STTran.forward lib/sttran.py:375-411.  `stream` is a hipStream_t (NULL = default stream). */
int sttran_forward(SttranHandle* h, const SttranInputs* in, const SttranOutputs* out, void* stream);
/* LANES (no reference counterpart): the reference's loop forwards ONE clip per call (tools/test_STTran.py:81-84,
 * dataloader/wk_action_genome.py:622-627), and one 176-pair clip cannot fill 256 CUs: every launch's ramp / prologue /
 * epilogue runs with the matrix pipes idle.  A handle can own up to STTRAN_MAX_LANES lanes: each lane has its own
 * workspace, stream-K park space, index staging and HIP stream (the weights are shared), so consecutive calls on
 * different lanes overlap on the device and one call's fixed costs hide under another call's MFMAs.
 *   sttran_set_lanes(h, n)            n lanes (synchronises the device; the default is 1)
 *   sttran_forward_lane(h, l, in, out, stream)
 *                                      the forward of sttran_forward on lane l's OWN stream, forked from `stream` with an
 *                                      event (everything enqueued on `stream` so far -- the producer of the entry --
 *                                      precedes it); `stream` does NOT wait for the result
 *   sttran_lane_join(h, l, stream)    `stream` waits for lane l's last forward (l = -1: for every lane's): the consumer
 *                                      of the outputs calls it first; sttran_sync_check joins all lanes itself
 *   sttran_lane_stream(h, l, &s)      lane l's hipStream_t (a framework allocator that frees per-stream needs it)
 * Results are bit-identical to sttran_forward's.  Calls on one handle stay serialised on the HOST by the caller; a lane
 * is reused only by a later call on the same lane, which its stream orders behind the earlier one.  Inputs and outputs
 * of a lane call must stay alive until that lane has been joined. */
#define STTRAN_MAX_LANES 8
int sttran_set_lanes(SttranHandle* h, int32_t lanes);
int32_t sttran_num_lanes(SttranHandle* h);
int sttran_forward_lane(SttranHandle* h, int32_t lane, const SttranInputs* in, const SttranOutputs* out, void* stream);
int sttran_lane_join(SttranHandle* h, int32_t lane, void* stream);
int sttran_lane_stream(SttranHandle* h, int32_t lane, void** stream);
/* Waits for `stream` and reports index errors the kernels met (pair_idx / labels out of range:
 * torch would raise an IndexError at lib/sttran.py:381-393; the kernels clamp and flag). */
int sttran_sync_check(SttranHandle* h, void* stream);
/* `del model` */
void sttran_destroy(SttranHandle* h);
const char* sttran_last_error(SttranHandle* h);
const char* sttran_version(void);

/* The step immediately before the path: union box and the two soft box masks of every
 * (subject, object) pair -- `union_boxes` and `draw_union_boxes(pair_rois, 27) - 0.5` of
 * lib/object_detector.py:110-124 / lib/draw_rectangles/draw_rectangles.pyx:27-67, on the device
 * instead of a D2H copy + Cython loop + H2D copy.  boxes [B,5] (col 0 = frame id), pair_idx [P,2],
 * im_idx [P] float (may be NULL), union_boxes [P,5] (may be NULL), spatial_masks [P,2,pool,pool]. */
int sttran_union_boxes_masks(const float* boxes, const int64_t* pair_idx, const float* im_idx, int64_t num_pairs,
                             int32_t pool, float* union_boxes, float* spatial_masks, void* stream);

/* The step immediately after the path: Recall@K matching of one clip's predictions with
 * its ground truth -- the per-frame core of `SceneGraphEvaluator.evaluate_scene_graph`
 * (lib/evaluation_recall.py:397-465): candidate selection of the three metrics (:209-235 with graph
 * constraint, :321-350 no constraint, :257-300 semi constraint), ordering by
 * subj_score * obj_score * predicate_score (:630-695) and matching by class triple + two float64
 * IoUs >= iou_threshold (:731-773, lib/fpn/box_intersections_cpu/bbox.pyx:21-61).
 * All pointers are device pointers.  Predictions are sttran_forward's outputs (attention = logits, the
 * softmax of :400 is applied here).  Ground truth is packed per clip: frame f owns ground-truth boxes
 * gt_box_off[f]..gt_box_off[f+1] (box 0 = the person) and relations gt_rel_off[f]..gt_rel_off[f+1];
 * gt_rels rows are (subject box, object box, predicate id) with box ids local to the frame.
 * flags [num_gt_rels, 9] uint8: flags[g][3*m + k] = 1 when relation g is hit within the first
 * {10, 20, 50}[k] predictions of metric m (0 with constraint, 1 no constraint, 2 semi constraint).
 * status (int32, device, caller zeroes it): bit 0 = unused since round 3 (a frame of any size is scored: sttran_eval_max_pairs),
 * bit 1 = pair_idx out of range; frames that set a bit get all-zero flags. */
typedef struct SttranEvalInputs {
  int32_t struct_size;
  int32_t num_frames;            /* frames in the ground truth (frame ids 0..num_frames-1)        */
  int32_t num_pairs;             /* P                                                              */
  int32_t num_boxes;             /* B                                                              */
  int32_t num_gt_rels;           /* rows of gt_rels / flags                                        */
  int32_t attention_classes, spatial_classes, contact_classes;   /* 3, 6, 17                      */
  int32_t im_idx_dtype;          /* STTRAN_DTYPE_F32 or STTRAN_DTYPE_I64                           */
  int32_t reserved;
  double iou_threshold;          /* 0.5 (tools/test_STTran.py:69)                                   */
  const float* attention_logits; /* [P, attention_classes]                                         */
  const float* spatial;          /* [P, spatial_classes]  probabilities                            */
  const float* contacting;       /* [P, contact_classes]  probabilities                            */
  const int64_t* pair_idx;       /* [P, 2]                                                         */
  const void* im_idx;            /* [P] ascending frame id                                         */
  const float* boxes;            /* [B, 5] col 0 = frame id                                        */
  const int64_t* classes;        /* [B] labels (predcls) or pred_labels                            */
  const float* obj_scores;       /* [B] scores (predcls) or pred_scores                            */
  const int32_t* gt_box_off;     /* [num_frames + 1]                                               */
  const float* gt_boxes;         /* [G, 4] x1,y1,x2,y2 (float32, as the reference rounds them :757) */
  const int32_t* gt_classes;     /* [G]                                                            */
  const int32_t* gt_rel_off;     /* [num_frames + 1]                                               */
  const int32_t* gt_rels;        /* [num_gt_rels, 3]                                               */
} SttranEvalInputs;
int sttran_eval_recall(const SttranEvalInputs* in, uint8_t* flags, int32_t* status, void* stream);
/* pairs of one frame that fit ONE pass of the kernel's key buffer for the given number of predicate columns (26 -> 96);
 * larger frames are scored in several passes whose top-50 lists are merged: not a limit, a performance knee */
int32_t sttran_eval_max_pairs(int32_t num_predicates);

/* SGDet WITHOUT weak supervision: the `else` branch of `ObjectClassifier.forward`, lib/sttran.py:185-283,
 * i.e. what runs in front of the relation transformer when `STTran(mode='sgdet', is_wks=False)` is in eval mode.
 *
 * sttran_objcls_select: `clean_class` for classes 5, 8, 17 (:52-85,197-199); per frame and per class (arg-max of the
 * distribution) greedy NMS at `nms_threshold` in descending score order (:203-237; the `nms` op is
 * fasterRCNN/lib/model/csrc/cuda/nms.cu:13-131 -- IoU with +1 pixel extents, suppression on IoU > threshold;
 * `nms_ge` != 0 selects the >= of cpu/nms_cpu.cpp:62); `pred_scores` / `pred_labels` = max / arg-max + 2 over columns
 * 1.. (:243-244); `human_idx` = per frame the row with the largest column-0 score, 0 for a frame without boxes, whose
 * label becomes 1 and whose score becomes that column-0 value (:247-254, the empty-frame assignment to row 0 included);
 * pairs (human of the frame, every other-labelled box of the frame) in frame order (:256-268).
 * Inputs are device pointers; `boxes` [B,5] must be sorted by frame id (column 0) with every id in [0, num_frames):
 * STTRAN_ERR_ORDER otherwise (the reference selects rows by `boxes[:,0] == i` and takes any order; sort first).  Every
 * output array must hold `capacity` >= 4 * num_boxes rows (a box has at most one copy per clean_class pass, and only the
 * newest copy of a chain can be copied again: 1 + 3; NMS only removes).
 * The output sizes are data dependent: the call synchronises the stream ONCE and returns them through
 * *num_boxes_out / *num_pairs_out (the reference synchronises dozens of times in this branch).
 * `out_source_row` (optional) tells which input row each output box is a copy of. */
typedef struct SttranObjclsSelect {
  uint32_t struct_size;          /* = sizeof(SttranObjclsSelect) */
  int32_t num_frames;            /* b = int(boxes[-1, 0] + 1)                         */
  int64_t num_boxes;             /* B                                                 */
  int32_t num_cols;              /* columns of `distribution` = len(obj_classes) - 1  */
  int32_t feat_dim;              /* columns of `features` (0 with features == NULL)   */
  float nms_threshold;           /* 0.6 (lib/sttran.py:227)                           */
  int32_t nms_ge;                /* 0: IoU > thr (nms.cu, what the reference runs); 1: IoU >= thr (nms_cpu.cpp) */
  const float* boxes;            /* [B,5]                 entry['boxes']              */
  const float* distribution;     /* [B,num_cols]          entry['distribution']       */
  const float* features;         /* [B,feat_dim] or NULL  entry['features']           */
  const int64_t* pred_labels;    /* [B]                   entry['pred_labels'] (the detector's) */
  int64_t capacity;              /* rows of every out_* array, >= 4 * num_boxes       */
  float* out_boxes;              /* [capacity,5]                                      */
  float* out_distribution;       /* [capacity,num_cols]                               */
  float* out_features;           /* [capacity,feat_dim] or NULL                       */
  float* out_pred_scores;        /* [capacity]                                        */
  int64_t* out_pred_labels;      /* [capacity]                                        */
  int32_t* out_source_row;       /* [capacity] or NULL                                */
  int64_t* out_pair_idx;         /* [capacity,2]                                      */
  float* out_im_idx;             /* [capacity]                                        */
  int64_t* out_human_idx;        /* [num_frames]                                      */
  void* scratch;                 /* sttran_objcls_scratch_bytes(num_boxes, num_frames) bytes of device memory */
  int64_t scratch_bytes;
} SttranObjclsSelect;
int64_t sttran_objcls_scratch_bytes(int64_t num_boxes, int32_t num_frames);
int sttran_objcls_select(const SttranObjclsSelect* args, int64_t* num_boxes_out, int64_t* num_pairs_out, void* stream);
/* `RCNN_roi_align(entry['fmaps'], union_boxes)`, lib/sttran.py:36,275: ROIAlign forward of
 * fasterRCNN/lib/model/csrc/cuda/ROIAlign_cuda.cu:65-118 (no coordinate rounding, adaptive sampling grid when
 * sampling_ratio <= 0, bilinear samples averaged per bin).  fmaps [T,C,H,W], rois [P,5] (frame index, x1,y1,x2,y2),
 * out [P,C,pooled,pooled]; device pointers; enqueue only. */
int sttran_roi_align(const float* fmaps, int32_t T, int32_t C, int32_t H, int32_t W, const float* rois, int64_t num_rois,
                     int32_t pooled, float spatial_scale, int32_t sampling_ratio, float* out, void* stream);

/* profiling (no reference counterpart; SURVEY 5 "Tracing / profiling: none") */
int sttran_profile_enable(SttranHandle* h, int32_t enable);
int sttran_profile_reset(SttranHandle* h);
int sttran_profile_read(SttranHandle* h, SttranProfile* out); /* synchronises the stream */
/* The same measurements broken down by (kernel template, problem shape): one entry per distinct launch site shape
 * since profile_reset, so that a per-kernel roofline can be recomputed from the bench line alone.  `ms` includes the
 * stream-K fix-up launch that belongs to a GEMM.  Call after sttran_profile_read (which folds the pending events in).
 * Writes min(count, cap) entries and the total count. */
typedef struct SttranProfEntry {
  char kernel[96];      /* e.g. "gemm_sk_kernel<GemmTile<256,128,4,2,B_KMAJOR_PAD>,EpiLinear>"  */
  int32_t cls;          /* STTRAN_PROF_*                                                         */
  int32_t reserved;
  int64_t M, N, K;      /* GEMM-shaped launches: C[M,N] += A[M,K] B[N,K]^T; otherwise rows / dim / 0 */
  uint64_t launches;
  double ms;            /* summed duration (HIP events on the forward's stream)                  */
  double flops;         /* summed algorithmic FLOPs                                              */
} SttranProfEntry;
int sttran_profile_entries(SttranHandle* h, SttranProfEntry* out, int32_t cap, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* STTRAN_HIP_H */
