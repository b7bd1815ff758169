"""Host-side mirror of the reference `lib/dsg_detr.py::STTran` (the DSG-DETR variant) on the shared HIP
kernels: same fusion front-end and relation heads as STTran, a stock post-norm encoder layer per frame,
then three encoder layers over per-object-class sequences with a sinusoidal frame encoding
(`lib/dsg_detr.py:514-572`).  Constructor as the reference (`:466-467`).  Only `mode='sgdet'` exists:
the reference's predcls branch does not run."""
from __future__ import annotations

from .. import _native as nat
from .sttran import STTran as _Base


class STTran(_Base):
    _model = nat.MODEL_DSG_DETR

    def __init__(self, mode="sgdet", attention_class_num=None, spatial_class_num=None, contact_class_num=None,
                 obj_classes=None):
        if mode != "sgdet":
            raise NotImplementedError("DSG-DETR: only the sgdet branch of the reference is runnable "
                                      "(lib/dsg_detr.py:183 feeds 2376-d features into Linear(2048, 512))")
        super().__init__(mode=mode, attention_class_num=attention_class_num, spatial_class_num=spatial_class_num,
                         contact_class_num=contact_class_num, obj_classes=obj_classes, enc_layer_num=1,
                         dec_layer_num=3, transformer_mode=None, is_wks=True, feat_dim=2048)


DSGDETR = STTran
