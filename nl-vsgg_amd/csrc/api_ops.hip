// api_ops.hip -- entry points that need no handle: union boxes + masks, the evaluator's matching core (8f-3),
// ObjectClassifier box / pair selection and ROIAlign (8f-2).
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

extern "C" {

int sttran_union_boxes_masks(const float* boxes, const int64_t* pair_idx, const float* im_idx, int64_t num_pairs,
                             int32_t pool, float* union_boxes, float* spatial_masks, void* stream) {
  if (!boxes || !pair_idx || !spatial_masks || num_pairs < 0 || pool <= 0 || pool > 64) return STTRAN_ERR_INVALID;
  return launch_union_boxes_masks(reinterpret_cast<hipStream_t>(stream), boxes, pair_idx, im_idx, (int)num_pairs, pool,
                                  union_boxes, spatial_masks) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_eval_recall(const SttranEvalInputs* in, uint8_t* flags, int32_t* status, void* stream) {
  if (!in || in->struct_size != (int32_t)sizeof(SttranEvalInputs) || !status) return STTRAN_ERR_INVALID;
  if (in->num_frames < 0 || in->num_pairs < 0 || in->num_boxes < 0 || in->num_gt_rels < 0) return STTRAN_ERR_INVALID;
  const int ncol = in->attention_classes + in->spatial_classes + in->contact_classes;
  // the semi-constraint rule reads columns 0,1 / 3,4 / 9,10 (lib/evaluation_recall.py:270-276)
  if (in->attention_classes < 2 || in->spatial_classes < 1 || in->contact_classes < 1 || ncol < 11 || ncol > 32)
    return STTRAN_ERR_INVALID;
  if (in->im_idx_dtype != STTRAN_DTYPE_F32 && in->im_idx_dtype != STTRAN_DTYPE_I64) return STTRAN_ERR_INVALID;
  if (in->num_frames == 0 || in->num_gt_rels == 0) return STTRAN_OK;
  if (!flags || !in->gt_box_off || !in->gt_boxes || !in->gt_classes || !in->gt_rel_off || !in->gt_rels)
    return STTRAN_ERR_INVALID;
  if (in->num_pairs > 0 && (!in->attention_logits || !in->spatial || !in->contacting || !in->pair_idx || !in->im_idx ||
                            !in->boxes || !in->classes || !in->obj_scores))
    return STTRAN_ERR_INVALID;
  hipError_t err = launch_eval_recall(reinterpret_cast<hipStream_t>(stream), in->attention_logits, in->spatial,
                                      in->contacting, in->pair_idx, in->im_idx, in->im_idx_dtype == STTRAN_DTYPE_I64,
                                      in->boxes, in->classes, in->obj_scores, in->num_pairs, in->num_boxes,
                                      in->attention_classes, in->spatial_classes, in->contact_classes, in->num_frames,
                                      in->gt_box_off, in->gt_boxes, in->gt_classes, in->gt_rel_off, in->gt_rels,
                                      in->iou_threshold, flags, status);
  return err == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int32_t sttran_eval_max_pairs(int32_t num_predicates) { return eval_max_pairs_per_frame(num_predicates); }

int64_t sttran_objcls_scratch_bytes(int64_t num_boxes, int32_t num_frames) {
  if (num_boxes < 0 || num_frames < 0) return 0;
  return (int64_t)objcls_scratch_bytes(num_boxes, num_frames);
}

int sttran_objcls_select(const SttranObjclsSelect* a, int64_t* num_boxes_out, int64_t* num_pairs_out, void* stream) {
  if (!a || a->struct_size != sizeof(SttranObjclsSelect) || !num_boxes_out || !num_pairs_out) return STTRAN_ERR_INVALID;
  if (a->num_boxes <= 0 || a->num_frames <= 0) return STTRAN_ERR_EMPTY;
  if (a->num_boxes > (1 << 26) || a->num_cols < 2 || a->num_cols > 64 || a->feat_dim < 0 || a->capacity < 4 * a->num_boxes)
    return STTRAN_ERR_INVALID;
  if (!a->boxes || !a->distribution || !a->pred_labels || !a->out_boxes || !a->out_distribution || !a->out_pred_scores ||
      !a->out_pred_labels || !a->out_pair_idx || !a->out_im_idx || !a->out_human_idx || !a->scratch ||
      (a->features != nullptr) != (a->out_features != nullptr) || (a->features && a->feat_dim <= 0) ||
      a->scratch_bytes < (int64_t)objcls_scratch_bytes(a->num_boxes, a->num_frames))
    return STTRAN_ERR_INVALID;
  int32_t host[3] = {0, 0, 0};
  hipError_t e = launch_objcls_select(reinterpret_cast<hipStream_t>(stream), a->boxes, a->distribution, a->features, a->pred_labels,
                                      a->num_boxes, a->num_frames, a->num_cols, a->feat_dim, a->nms_threshold, a->nms_ge, a->capacity,
                                      a->out_boxes, a->out_distribution, a->out_features, a->out_pred_scores, a->out_pred_labels,
                                      a->out_source_row, a->out_pair_idx, a->out_im_idx, a->out_human_idx, a->scratch, host);
  if (e != hipSuccess) return STTRAN_ERR_HIP;
  if (host[2] & 2) return STTRAN_ERR_ORDER;        // boxes not sorted by frame id, or a frame id outside [0, num_frames)
  *num_boxes_out = host[0];
  *num_pairs_out = host[1];
  return STTRAN_OK;
}

int sttran_roi_align(const float* fmaps, int32_t T, int32_t C, int32_t H, int32_t W, const float* rois, int64_t num_rois,
                     int32_t pooled, float spatial_scale, int32_t sampling_ratio, float* out, void* stream) {
  if (!fmaps || T <= 0 || C <= 0 || H <= 0 || W <= 0 || num_rois < 0 || pooled <= 0 || pooled > 64 || (num_rois > 0 && (!rois || !out)))
    return STTRAN_ERR_INVALID;
  return launch_roi_align(reinterpret_cast<hipStream_t>(stream), fmaps, T, C, H, W, rois, num_rois, pooled, spatial_scale,
                          sampling_ratio, out) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}


}  // extern "C"
