"""ctypes binding of include/sttran_hip.h (libsttran_hip.so).

There is no CPU fallback: if the HIP library is missing or does not load, importing the product
path raises -- see `load()`.  `import torch` must happen before the library is loaded so that it
binds to the same libamdhip64 (identical soname) PyTorch-ROCm already mapped.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsttran_hip.so")
# A/B runs of tools/ only (an experiment build, or the previous commit's library next to the current one): another build
# of the SAME library, e.g. STTRAN_LIB=nl-vsgg_amd/csrc/ab/libsttran_hip_A.so.  Still this package's HIP library -- there
# is no other implementation to select.
if os.environ.get("STTRAN_LIB"):
    LIB_PATH = os.path.abspath(os.environ["STTRAN_LIB"])

STTRAN_OK = 0
STTRAN_ERR_INVALID = 1
ERR_NAMES = {1: "INVALID", 2: "HIP", 3: "EMPTY", 4: "WEIGHTS", 5: "ORDER", 6: "LIMIT", 7: "INDEX"}
MODE = {"predcls": 0, "sgcls": 1, "sgdet": 2}
MODEL_STTRAN, MODEL_DSG_DETR = 0, 1
DTYPE_F32, DTYPE_I64, DTYPE_I32 = 0, 1, 2
PROF_CLASSES = 8
PROF_NAMES = ["gemm", "union_conv", "mask_conv", "attention", "layernorm", "index", "other", "_"]


class SttranConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("mode", C.c_int32),
                ("enc_layers", C.c_int32), ("dec_layers", C.c_int32),
                ("attention_classes", C.c_int32), ("spatial_classes", C.c_int32),
                ("contact_classes", C.c_int32), ("num_obj_classes", C.c_int32),
                ("feat_dim", C.c_int32), ("embed_dim", C.c_int32), ("nhead", C.c_int32),
                ("ffn_dim", C.c_int32), ("model", C.c_int32)]


class SttranInputs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("num_clips", C.c_int32),
                ("num_boxes", C.c_int64), ("num_pairs", C.c_int64),
                ("num_frames", C.c_int32), ("im_idx_dtype", C.c_int32),
                ("clip_num_frames", C.POINTER(C.c_int32)), ("frame_counts", C.POINTER(C.c_int32)),
                ("features", C.c_void_p), ("pair_idx", C.c_void_p), ("labels", C.c_void_p),
                ("union_feat", C.c_void_p), ("spatial_masks", C.c_void_p), ("im_idx", C.c_void_p),
                ("boxes", C.c_void_p), ("distribution", C.c_void_p),
                # per-clip pointer tables (host arrays of device pointers), include/sttran_hip.h form (2)
                ("clip_features", C.POINTER(C.c_void_p)), ("clip_pair_idx", C.POINTER(C.c_void_p)),
                ("clip_labels", C.POINTER(C.c_void_p)), ("clip_union_feat", C.POINTER(C.c_void_p)),
                ("clip_spatial_masks", C.POINTER(C.c_void_p)), ("clip_boxes", C.POINTER(C.c_void_p)),
                ("clip_distribution", C.POINTER(C.c_void_p)),
                ("clip_num_boxes", C.POINTER(C.c_int64)), ("clip_num_pairs", C.POINTER(C.c_int64))]


INPUTS_V1_SIZE = 112       # STTRAN_INPUTS_V1_SIZE: the struct without the pointer tables (round-2 callers)


class SttranOutputs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("attention_distribution", C.c_void_p), ("spatial_distribution", C.c_void_p),
                ("contacting_distribution", C.c_void_p), ("distribution", C.c_void_p),
                ("rel_features_tap", C.c_void_p), ("local_output_tap", C.c_void_p),
                ("global_output_tap", C.c_void_p)]


class SttranProfile(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("forwards", C.c_uint32),
                ("ms", C.c_double * PROF_CLASSES), ("flops", C.c_double * PROF_CLASSES),
                ("bytes", C.c_double * PROF_CLASSES), ("launches", C.c_uint64 * PROF_CLASSES)]


class SttranProfEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 96), ("cls", C.c_int32), ("reserved", C.c_int32),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64), ("launches", C.c_uint64),
                ("ms", C.c_double), ("flops", C.c_double)]


class SttranEvalInputs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("num_frames", C.c_int32), ("num_pairs", C.c_int32),
                ("num_boxes", C.c_int32), ("num_gt_rels", C.c_int32),
                ("attention_classes", C.c_int32), ("spatial_classes", C.c_int32), ("contact_classes", C.c_int32),
                ("im_idx_dtype", C.c_int32), ("reserved", C.c_int32), ("iou_threshold", C.c_double),
                ("attention_logits", C.c_void_p), ("spatial", C.c_void_p), ("contacting", C.c_void_p),
                ("pair_idx", C.c_void_p), ("im_idx", C.c_void_p), ("boxes", C.c_void_p),
                ("classes", C.c_void_p), ("obj_scores", C.c_void_p),
                ("gt_box_off", C.c_void_p), ("gt_boxes", C.c_void_p), ("gt_classes", C.c_void_p),
                ("gt_rel_off", C.c_void_p), ("gt_rels", C.c_void_p)]


class SttranObjclsSelect(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("num_frames", C.c_int32), ("num_boxes", C.c_int64),
                ("num_cols", C.c_int32), ("feat_dim", C.c_int32), ("nms_threshold", C.c_float), ("nms_ge", C.c_int32),
                ("boxes", C.c_void_p), ("distribution", C.c_void_p), ("features", C.c_void_p), ("pred_labels", C.c_void_p),
                ("capacity", C.c_int64), ("out_boxes", C.c_void_p), ("out_distribution", C.c_void_p),
                ("out_features", C.c_void_p), ("out_pred_scores", C.c_void_p), ("out_pred_labels", C.c_void_p),
                ("out_source_row", C.c_void_p), ("out_pair_idx", C.c_void_p), ("out_im_idx", C.c_void_p),
                ("out_human_idx", C.c_void_p), ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64)]


# every symbol include/sttran_hip.h (the drop-in boundary) and include/sttran_hip_debug.h (test hooks, experiment
# switches: names `sttran_debug_*` and sttran_set_gemm_engine) declare: (name, restype, argtypes)
SYMBOLS = [
    ("sttran_create", C.c_int, [C.POINTER(SttranConfig), C.POINTER(C.c_void_p)]),
    ("sttran_load_tensor", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64),
                                     C.c_int32, C.c_int32, C.c_int32]),
    ("sttran_finalize_weights", C.c_int, [C.c_void_p]),
    ("sttran_missing_keys", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("sttran_reserve", C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    ("sttran_set_gemm_engine", C.c_int, [C.c_void_p, C.c_int32]),
    ("sttran_forward", C.c_int, [C.c_void_p, C.POINTER(SttranInputs), C.POINTER(SttranOutputs), C.c_void_p]),
    ("sttran_set_lanes", C.c_int, [C.c_void_p, C.c_int32]),
    ("sttran_num_lanes", C.c_int32, [C.c_void_p]),
    ("sttran_forward_lane", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(SttranInputs), C.POINTER(SttranOutputs), C.c_void_p]),
    ("sttran_lane_join", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    ("sttran_lane_stream", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("sttran_sync_check", C.c_int, [C.c_void_p, C.c_void_p]),
    ("sttran_destroy", None, [C.c_void_p]),
    ("sttran_last_error", C.c_char_p, [C.c_void_p]),
    ("sttran_version", C.c_char_p, []),
    ("sttran_profile_enable", C.c_int, [C.c_void_p, C.c_int32]),
    ("sttran_profile_reset", C.c_int, [C.c_void_p]),
    ("sttran_profile_read", C.c_int, [C.c_void_p, C.POINTER(SttranProfile)]),
    ("sttran_profile_entries", C.c_int, [C.c_void_p, C.POINTER(SttranProfEntry), C.c_int32, C.POINTER(C.c_int32)]),
    ("sttran_union_boxes_masks", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    ("sttran_eval_recall", C.c_int, [C.POINTER(SttranEvalInputs), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sttran_eval_max_pairs", C.c_int32, [C.c_int32]),
    ("sttran_objcls_scratch_bytes", C.c_int64, [C.c_int64, C.c_int32]),
    ("sttran_objcls_select", C.c_int, [C.POINTER(SttranObjclsSelect), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    ("sttran_roi_align", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    ("sttran_debug_gemm", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("sttran_debug_gemm_padded", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    ("sttran_debug_gemm_emulated", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    ("sttran_debug_gemm_emulated_t16", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    ("sttran_debug_x3t16_bench", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    ("sttran_debug_mfma_peak", C.c_int, [C.c_int32, C.POINTER(C.c_double)]),
    ("sttran_debug_plan_tile", C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    ("sttran_debug_guarded_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("sttran_debug_guarded_free", C.c_int, [C.c_void_p]),
    ("sttran_debug_guarded_return_addresses", C.c_int, [C.c_int32]),
    ("sttran_debug_mask_conv1_pool", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_void_p]),
    ("sttran_debug_layernorm", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                         C.c_void_p]),
    ("sttran_debug_attention", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    ("sttran_debug_dsg_layout", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    ("sttran_debug_attention_classes", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_int32, C.c_int32, C.c_void_p]),
]

_lib = None


class SttranError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sttran error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class SttranIndexError(SttranError, IndexError):
    """pair_idx / labels out of range: what torch raises as an IndexError at lib/sttran.py:381-393."""


def load():
    """Load libsttran_hip.so and bind every declared symbol.  Raises if the library is absent:
    the product path has no other implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            f"(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C nl-vsgg_amd/csrc`). "
            f"There is no CPU fallback for the STTran hot path.")
    import torch  # noqa: F401  (maps PyTorch-ROCm's libamdhip64 first; see module docstring)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, handle, rc):
    if rc != STTRAN_OK:
        msg = lib.sttran_last_error(handle) if handle else b""
        raise SttranError(rc, (msg or b"").decode("utf-8", "replace"))
