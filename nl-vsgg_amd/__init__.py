"""nl-vsgg_amd: MI355X-native STTran relation-transformer hot path of rlqja1107/NL-VSGG.

Only the path named by BASELINE.json's north_star lives here:
  csrc/   hand-written HIP (gfx950) kernels + the C-ABI (include/sttran_hip.h)
  lib/    host-side mirror of the reference interface (`lib/sttran.py::STTran`,
          `lib/evaluation_recall.py::SceneGraphEvaluator`) and synthetic-data helpers
"""
__version__ = "0.1.0"
