"""No kernel of the path touches memory past a caller's buffer: GEMM operands and whole forwards run on tensors whose
last element is the last MAPPED byte of its address range (sttran_debug_guarded_alloc), in a child process -- an
out-of-bounds access is a GPU memory fault that aborts the child.  (Round 3 shipped such a read for one commit: a lane
whose rows were both past M fetched its residual operand from row >= M; every parity test passed, the default bench
faulted on the 64x36 clip.)"""
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "guarded_child.py")


def _run(what, timeout=600, env=None):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, CHILD, what], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    if "NOVMM" in r.stdout:
        pytest.skip("the driver has no virtual-memory API: " + r.stdout.strip())
    assert r.returncode == 0 and f"OK {what}" in r.stdout, f"rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def test_guarded_allocator_itself():
    _run("probe")


def test_gemm_operands_at_the_end_of_a_mapping():
    """A [M, ceil32(K)], W, bias, residual and C exactly as large as the contract says (no slack row), every tile"""
    _run("gemm")


def test_forward_inputs_at_the_end_of_a_mapping():
    """predcls and sgdet entries (ragged, empty frames, 16x12, 35 pairs per frame) and the by-pointer batch of them:
    same bits as with ordinary tensors"""
    _run("forward")


def test_other_entry_points_on_guarded_inputs():
    """DSG-DETR forward, detector-output selection + ROIAlign (f-2), union boxes / masks (f-1), the device evaluator (f-3)"""
    _run("aux")


def test_library_workspace_and_weights_at_the_end_of_their_mappings():
    """STTRAN_GUARD_WORKSPACE=1: every internal buffer (activations, QKV, parked stream-K partials, index maps) and every
    weight tensor is a guarded allocation; STTran predcls / sgdet (64x36, ragged, empty frames, a by-pointer batch on a
    fresh handle, the bf16x3 engine) and DSG-DETR produce the same bits as with ordinary allocations"""
    _run("probe")
    digest = lambda out: [l for l in out.splitlines() if l.startswith("OK workspace")][0].split()[-1]
    a = digest(_run("workspace", env={"STTRAN_GUARD_WORKSPACE": "1"}))
    b = digest(_run("workspace", env={"STTRAN_GUARD_WORKSPACE": "0"}))
    assert a == b
