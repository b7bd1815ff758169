"""CPU-only checks of the C-ABI boundary: the library is built, loads without a GPU, and exports
exactly the entry points include/sttran_hip.h declares, with struct sizes the binding agrees on."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as ge
    from nl_vsgg_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        ge.build()
    return _native


def _declared(header="sttran_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sttran_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(native):
    """every entry point of the drop-in boundary (sttran_hip.h) and of the lab header (sttran_hip_debug.h) is exported
    and bound; the boundary carries no test hook and no experiment switch"""
    lib = native.load()
    names, lab = _declared(), _declared("sttran_hip_debug.h")
    assert len(names) >= 20 and len(lab) >= 10
    for n in names + lab:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert sorted(s[0] for s in native.SYMBOLS) == sorted(names + lab)
    assert not [n for n in names if "debug" in n or "engine" in n], "lab symbols in the product header"
    assert all(n.startswith("sttran_debug_") or n == "sttran_set_gemm_engine" for n in lab)
    for n in ("sttran_set_lanes", "sttran_forward_lane", "sttran_lane_join", "sttran_lane_stream", "sttran_num_lanes"):
        assert n in names


def test_version_and_null_handle(native):
    lib = native.load()
    assert b"gfx950" in lib.sttran_version()
    assert lib.sttran_last_error(None) == b"null handle"
    assert lib.sttran_finalize_weights(None) == 1


def test_struct_sizes(native):
    # the library rejects any struct whose struct_size differs from its own sizeof
    assert C.sizeof(native.SttranConfig) == 14 * 4
    assert native.INPUTS_V1_SIZE == 8 + 16 + 8 + 16 + 8 * 8                 # round-2 callers' struct (no pointer tables)
    assert C.sizeof(native.SttranInputs) == native.INPUTS_V1_SIZE + 9 * 8    # + 7 pointer tables + 2 size arrays
    assert native.SttranInputs.clip_features.offset == native.INPUTS_V1_SIZE
    assert C.sizeof(native.SttranOutputs) == 8 + 7 * 8
    assert C.sizeof(native.SttranProfile) == 8 + 4 * 8 * 8
    lib = native.load()
    cfg = native.SttranConfig(struct_size=4)
    h = C.c_void_p()
    assert lib.sttran_create(C.byref(cfg), C.byref(h)) == 1      # STTRAN_ERR_INVALID, no GPU touched


def test_ctypes_structs_match_the_c_header(native, tmp_path):
    """what a C compiler makes of include/sttran_hip.h (plain C11: the header must not need C++) vs the ctypes mirror"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "sizes.c"
    src.write_text("""
#include <stddef.h>
#include <stdio.h>
#include "sttran_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %u %zu %zu\\n", sizeof(SttranConfig), sizeof(SttranInputs), sizeof(SttranOutputs),
         sizeof(SttranProfile), sizeof(SttranProfEntry), sizeof(SttranEvalInputs), sizeof(SttranObjclsSelect),
         STTRAN_INPUTS_V1_SIZE, offsetof(SttranInputs, clip_features), offsetof(SttranInputs, clip_num_pairs));
  return 0;
}
""")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(native.SttranConfig), C.sizeof(native.SttranInputs), C.sizeof(native.SttranOutputs),
            C.sizeof(native.SttranProfile), C.sizeof(native.SttranProfEntry), C.sizeof(native.SttranEvalInputs),
            C.sizeof(native.SttranObjclsSelect), native.INPUTS_V1_SIZE, native.SttranInputs.clip_features.offset,
            native.SttranInputs.clip_num_pairs.offset]
    assert got == want


def test_eval_recall_argument_checks(native):
    """sttran_eval_recall validates its struct before any HIP call (so this runs without a GPU)"""
    lib = native.load()
    assert C.sizeof(native.SttranEvalInputs) == 10 * 4 + 8 + 13 * 8
    assert lib.sttran_eval_max_pairs(26) == 96
    status = C.c_int32(0)
    inp = native.SttranEvalInputs(struct_size=4)
    assert lib.sttran_eval_recall(C.byref(inp), None, C.byref(status), None) == 1
    inp = native.SttranEvalInputs(struct_size=C.sizeof(native.SttranEvalInputs), num_frames=1, num_pairs=1, num_boxes=2,
                                  num_gt_rels=1, attention_classes=3, spatial_classes=2, contact_classes=2,
                                  iou_threshold=0.5)
    assert lib.sttran_eval_recall(C.byref(inp), None, C.byref(status), None) == 1      # fewer than 11 predicate columns
    inp.spatial_classes, inp.contact_classes = 6, 17
    assert lib.sttran_eval_recall(C.byref(inp), None, C.byref(status), None) == 1      # null buffers
    inp.num_frames = 0
    assert lib.sttran_eval_recall(C.byref(inp), None, C.byref(status), None) == 0      # nothing to do


def test_missing_library_fails_loudly(native, monkeypatch):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libsttran_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        native.load()
