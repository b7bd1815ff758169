"""Kernel-level parity through the C ABI test hooks (sttran_debug_*), one kernel class at a time.
The comparison values are float64 torch/numpy restatements of the single op (a fp32 reference of a
floating-point kernel, as the brief asks), tolerance 1e-4 relative to the output scale."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nl_vsgg_amd import _native
    return _native.load()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _gemm(lib, A, W, bias=None, res=None, rowidx=None, relu=0, tile=0, split=0):
    M = A.shape[0] if rowidx is None else rowidx.shape[0]
    N, K = W.shape
    Cc = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32)
    rc = lib.sttran_debug_gemm(_p(A), _p(rowidx), _p(W), _p(bias), _p(res), _p(Cc), M, N, K, relu, tile, split, None)
    assert rc == 0
    torch.cuda.synchronize()
    return Cc


def _ref(A, W, bias=None, res=None, rowidx=None, relu=0):
    a = A.double() if rowidx is None else A.double()[rowidx.long()]
    r = a @ W.double().T
    if bias is not None:
        r = r + bias.double()
    if relu:
        r = r.clamp_min(0)
    if res is not None:
        r = r + res.double()
    return r


SHAPES = [(1, 1, 4), (2, 3872, 1936), (33, 70, 100), (176, 512, 2048), (330, 5808, 1936), (257, 129, 36),
          (64, 64, 32), (300, 26, 1936), (130, 1936, 2048)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_gemm_tiles(lib, M, N, K, tile):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    out = _gemm(lib, A, W, bias=b, tile=tile, split=1)
    ref = _ref(A, W, b)
    tol = 1e-5 * (K ** 0.5) * 4 + 1e-5
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item() / (K ** 0.5))


@pytest.mark.parametrize("M,N,K", [(2, 3872, 1936), (33, 70, 100), (330, 5808, 1936), (257, 129, 36), (300, 26, 1936),
                                   (2816, 1936, 1936), (64, 64, 32), (5280, 1936, 2048), (200, 1024, 2376)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_gemm_padded_path(lib, M, N, K, tile):
    """the product path: rows padded to a multiple of 32 columns, W zero there, A holding arbitrary finite values there
    (they must not reach the result), no zero-select in the kernel, loads two K-steps ahead"""
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 5 * K + tile)
    Kp = (K + 31) // 32 * 32
    A = torch.randn(M + 1, Kp, device="cuda", generator=g) * 3          # pad columns: garbage; +1 row of slack
    W = torch.zeros(N, Kp, device="cuda")
    W[:, :K] = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    Cc = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(W), Kp, _p(b), _p(res), _p(Cc), M, N, K, 1, tile, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = _ref(A[:M, :K], W[:, :K], b, res, relu=1)
    tol = 1e-5 * (K ** 0.5) * 12 + 1e-5
    assert torch.isfinite(Cc).all()
    assert (Cc.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item() / (K ** 0.5))


@pytest.mark.parametrize("M,N,K", [(2816, 1936, 1936), (5280, 1936, 2048), (330, 5808, 1936), (700, 3872, 1936), (1, 176, 32),
                                   (129, 352, 100), (21120, 1936, 1936)])
@pytest.mark.parametrize("tile", [5])
def test_gemm_tile_128x176(lib, M, N, K, tile):
    """the 16x16x4-MFMA tile of the N = 1936 family (gemm_f32_t16.h): exact in N, two workgroups per CU, XOR-swizzled LDS;
    bias + ReLU + residual through the 16-byte epilogue, stream-K remainders through its own fix-up kernel"""
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 5 * K)
    Kp = (K + 31) // 32 * 32
    A = torch.randn(M + 1, Kp, device="cuda", generator=g) * 3          # pad columns: garbage (W is zero there)
    W = torch.zeros(N, Kp, device="cuda")
    W[:, :K] = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    Cc = torch.full((M, N), float("nan"), device="cuda")
    assert lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(W), Kp, _p(b), _p(res), _p(Cc), M, N, K, 1, tile, None) == 0
    torch.cuda.synchronize()
    ref = _ref(A[:M, :K], W[:, :K], b, res, relu=1)
    tol = 1e-5 * (K ** 0.5) * 12 + 1e-5
    assert torch.isfinite(Cc).all()
    assert (Cc.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item() / (K ** 0.5))
    # deterministic: the parked stream-K partials are summed in a fixed order
    C2 = torch.empty_like(Cc)
    assert lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(W), Kp, _p(b), _p(res), _p(C2), M, N, K, 1, tile, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(Cc, C2)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7])
def test_gemm_padded_path_gathered_rows(lib, tile):
    """A rows gathered through an index (subj/obj FC, last-decoder-layer row pruning), every tile"""
    g = torch.Generator(device="cuda").manual_seed(40 + tile)
    M, N, K, R = 700, 1936, 1936, 300
    Kp = (K + 31) // 32 * 32
    A = torch.randn(R + 1, Kp, device="cuda", generator=g)
    W = torch.zeros(N, Kp, device="cuda")
    W[:, :K] = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    idx = torch.randint(0, R, (M,), device="cuda", generator=g, dtype=torch.int32)
    Cc = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.sttran_debug_gemm_padded(_p(A), Kp, _p(idx), _p(W), Kp, _p(b), None, _p(Cc), M, N, K, 0, tile, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = _ref(A[:R, :K], W[:, :K], b, rowidx=idx)
    assert (Cc.double() - ref).abs().max().item() < 2e-4


def test_gemm_tiles_agree(lib):
    """same MFMA instruction, same k order inside every accumulator whatever the tile: two tiles differ only by where the
    stream-K schedule splits K (rounding-level agreement)"""
    g = torch.Generator(device="cuda").manual_seed(77)
    M, N, K = 256, 256, 1936
    Kp = (K + 31) // 32 * 32
    A = torch.randn(M + 1, Kp, device="cuda", generator=g)
    W = torch.zeros(N, Kp, device="cuda")
    W[:, :K] = torch.randn(N, K, device="cuda", generator=g)
    outs = []
    for tile in (3, 2):
        Cc = torch.empty(M, N, device="cuda")
        assert lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(W), Kp, None, None, _p(Cc), M, N, K, 0, tile, None) == 0
        outs.append(Cc)
    torch.cuda.synchronize()
    assert (outs[0] - outs[1]).abs().max().item() < 1e-4       # split points may differ: rounding-level agreement


@pytest.mark.parametrize("split", [2, 4, 8])
def test_gemm_split_k(lib, split):
    g = torch.Generator(device="cuda").manual_seed(split)
    M, N, K = 176, 512, 12544
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.01
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    out = _gemm(lib, A, W, bias=b, res=res, relu=1, tile=3, split=split)
    ref = _ref(A, W, b, res, relu=1)
    assert (out.double() - ref).abs().max().item() < 2e-4


def test_gemm_gather_rows(lib):
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(50, 2048, device="cuda", generator=g)
    W = torch.randn(512, 2048, device="cuda", generator=g) * 0.02
    idx = torch.randint(0, 50, (176,), device="cuda", generator=g, dtype=torch.int32)
    out = _gemm(lib, A, W, rowidx=idx)
    ref = _ref(A, W, rowidx=idx)
    assert (out.double() - ref).abs().max().item() < 1e-4


def test_gemm_auto_plan_matches(lib):
    g = torch.Generator(device="cuda").manual_seed(9)
    A = torch.randn(2240, 1936, device="cuda", generator=g)
    W = torch.randn(2048, 1936, device="cuda", generator=g) * 0.02
    out = _gemm(lib, A, W)
    assert (out.double() - _ref(A, W)).abs().max().item() < 2e-4


@pytest.mark.parametrize("rows,dim", [(1, 1936), (7, 1936), (330, 1936), (5, 2376), (3, 4096), (9, 8)])
def test_layernorm(lib, rows, dim):
    g = torch.Generator(device="cuda").manual_seed(rows + dim)
    x = torch.randn(rows, dim, device="cuda", generator=g) * 3 + 1
    gm = torch.rand(dim, device="cuda", generator=g) + 0.5
    bt = torch.rand(dim, device="cuda", generator=g) - 0.5
    y = torch.empty_like(x)
    assert lib.sttran_debug_layernorm(_p(x), _p(gm), _p(bt), _p(y), rows, dim, None) == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (dim,), gm.double(), bt.double(), 1e-5)
    assert (y.double() - ref).abs().max().item() < 1e-5


def _attn_ref(qkv, offs, lens, dim, nhead):
    hd = dim // nhead
    out = torch.zeros(qkv.shape[0], dim, dtype=torch.float64, device=qkv.device)
    q, k, v = qkv.double().split(dim, dim=1)
    for o, l in zip(offs, lens):
        if l == 0:
            continue
        qs = q[o:o + l].view(l, nhead, hd).transpose(0, 1) / hd ** 0.5
        ks = k[o:o + l].view(l, nhead, hd).transpose(0, 1)
        vs = v[o:o + l].view(l, nhead, hd).transpose(0, 1)
        a = torch.softmax(qs @ ks.transpose(1, 2), dim=-1)
        out[o:o + l] = (a @ vs).transpose(0, 1).reshape(l, dim)
    return out


@pytest.mark.parametrize("lens", [[1], [22] * 15, [70, 3, 35, 64, 33], [5, 0, 9], [129, 31], [480],
                                  # beyond one pass of the 480-key score block: chunks with a running softmax
                                  [481], [600, 7, 1000], [1537],
                                  # round 5: the short-sequence variant is chosen by length AND workgroup count -- launches that
                                  # fill the chip take the high-residency forms (<= 24 keys: <64,128,32,8>), 33..48 keys the
                                  # chunked <64,128,48,4>, 49..80 <32,64,80,3>; 25..32 keys stay on the all-loads-first form
                                  [11] * 300, [22] * 256, list(range(1, 25)) * 12, [24, 1, 0, 13] * 70, [32, 25, 7] * 100,
                                  [40, 33, 48, 2] * 8, [48] * 64, [49, 80, 64, 1] * 6, [77] * 40])
def test_attention(lib, lens):
    dim, nhead = 1936, 8
    g = torch.Generator(device="cuda").manual_seed(sum(lens))
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    tokens = int(sum(lens))
    qkv = torch.randn(tokens, 3 * dim, device="cuda", generator=g)
    qkv[:, :dim] *= 2.0          # sharpen the softmax a little
    out = torch.full((tokens, dim), float("nan"), device="cuda")
    so = torch.from_numpy(offs).cuda()
    sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_attention(_p(qkv), _p(so), _p(sl), len(lens), max(lens), _p(out), tokens, dim, nhead,
                                      None) == 0
    torch.cuda.synchronize()
    ref = _attn_ref(qkv, offs.tolist(), lens, dim, nhead)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 2e-5


def test_attention_spiked_scores(lib):
    """one key dominates one query by a large margin: the max-subtraction must hold (guide rule 26)."""
    dim, nhead, L = 1936, 8, 40
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(L, 3 * dim, device="cuda", generator=g)
    qkv[3, :dim] *= 40.0
    qkv[17, dim:2 * dim] = qkv[3, :dim] / 4
    out = torch.empty(L, dim, device="cuda")
    so = torch.zeros(1, dtype=torch.int32, device="cuda")
    sl = torch.full((1,), L, dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_attention(_p(qkv), _p(so), _p(sl), 1, L, _p(out), L, dim, nhead, None) == 0
    torch.cuda.synchronize()
    ref = _attn_ref(qkv, [0], [L], dim, nhead)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("spike_at", [5, 700, 1100])
def test_attention_spiked_scores_across_chunks(lib, spike_at):
    """1 200 keys = three passes of the score block: a key that dominates a query by a large margin sits in the first,
    second or third pass, so the running maximum jumps (and earlier partial sums are scaled down to ~0) or later
    passes contribute ~0 to a row whose maximum is already large"""
    dim, nhead, L = 1936, 8, 1200
    g = torch.Generator(device="cuda").manual_seed(spike_at)
    qkv = torch.randn(L, 3 * dim, device="cuda", generator=g)
    qkv[3, :dim] *= 40.0
    qkv[spike_at, dim:2 * dim] = qkv[3, :dim] / 4
    out = torch.empty(L, dim, device="cuda")
    so = torch.zeros(1, dtype=torch.int32, device="cuda")
    sl = torch.full((1,), L, dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_attention(_p(qkv), _p(so), _p(sl), 1, L, _p(out), L, dim, nhead, None) == 0
    torch.cuda.synchronize()
    ref = _attn_ref(qkv, [0], [L], dim, nhead)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,tile", [(330, 1936, 1936, 4), (176, 5808, 1936, 3), (2816, 1936, 1936, 1), (5280, 2048, 1936, 1),
                                        (700, 512, 12544, 2), (330, 1936, 2048, 0)])
def test_gemm_stream_k_reuse_stress(lib, M, N, K, tile):
    """stream-K partial tiles are parked in a slab that every launch reuses and summed by the fix-up launch: fresh
    operands on every launch, an fp64 reference, and three repeats that must agree bit for bit (fixed summation order;
    nothing from the previous problem may leak through the park space)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K + tile)
    Kp = (K + 31) // 32 * 32
    W = torch.zeros(N, Kp, device="cuda")
    b = torch.randn(N, device="cuda", generator=g)
    tol = 1e-5 * (K ** 0.5) * 12 + 1e-5
    for it in range(40):
        A = torch.randn(M + 1, Kp, device="cuda", generator=g) * (1.0 + it)       # scale changes: stale sums stand out
        W[:, :K] = torch.randn(N, K, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g)
        outs = []
        for rep in range(3):
            Cc = torch.full((M, N), float("nan"), device="cuda")
            assert lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(W), Kp, _p(b), _p(res), _p(Cc), M, N, K, 0, tile, None) == 0
            outs.append(Cc)
        torch.cuda.synchronize()
        ref = _ref(A[:M, :K], W[:, :K], b, res)
        err = (outs[0].double() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item() / (K ** 0.5)), (it, err)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), it


X3_HOOKS = ["sttran_debug_gemm_emulated", "sttran_debug_gemm_emulated_t16"]     # round 2's kernel / the 16x16x32 tiles of round 6


@pytest.mark.parametrize("hook,M,N,K", [(X3_HOOKS[0], 300, 512, 2048), (X3_HOOKS[0], 2816, 1936, 1936), (X3_HOOKS[0], 1, 4, 32),
                                        (X3_HOOKS[0], 257, 130, 100), (X3_HOOKS[0], 5280, 2048, 1936),
                                        (X3_HOOKS[1], 300, 512, 2048), (X3_HOOKS[1], 2816, 1936, 1936), (X3_HOOKS[1], 1, 176, 32),
                                        (X3_HOOKS[1], 257, 352, 100), (X3_HOOKS[1], 5280, 2048, 1936), (X3_HOOKS[1], 40000, 176, 64),
                                        (X3_HOOKS[1], 330, 5808, 1936), (X3_HOOKS[1], 1031, 128, 2376)])
def test_gemm_bf16x3_emulation_is_fp32_accurate(lib, hook, M, N, K):
    """the experimental bf16x3 engine (three bf16 planes per operand, six cross products, fp32 accumulate) must be as
    close to an fp64 reference as the exact-fp32 MFMA engine is -- it is an emulation of fp32, not a reduced precision"""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    Kp = (K + 31) // 32 * 32
    A = torch.randn(M + 1, Kp, device="cuda", generator=g) * 2
    A[:, K:] = 0                                              # the product keeps the pad columns zero
    W = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    C3 = torch.full((M, N), float("nan"), device="cuda")
    assert getattr(lib, hook)(_p(A), Kp, None, _p(W), K, _p(b), _p(res), _p(C3), M, N, K, 1, None) == 0
    Wp = torch.zeros(N, Kp, device="cuda"); Wp[:, :K] = W
    C1 = torch.full((M, N), float("nan"), device="cuda")
    assert lib.sttran_debug_gemm_padded(_p(A), Kp, None, _p(Wp), Kp, _p(b), _p(res), _p(C1), M, N, K, 1, 0, None) == 0
    torch.cuda.synchronize()
    ref = _ref(A[:M, :K], W, b, res, relu=1)
    e3 = (C3.double() - ref).abs().max().item()
    e1 = (C1.double() - ref).abs().max().item()
    assert torch.isfinite(C3).all()
    assert e3 <= 2.0 * e1 + 1e-6, (e3, e1)


@pytest.mark.parametrize("hook", X3_HOOKS)
@pytest.mark.parametrize("a_scale,w_scale", [(1e-30, 1e20), (1e20, 1e-30), (1e-20, 1e-10), (1e18, 1e18), (1e-3, 1e3)])
def test_gemm_bf16x3_operand_ranges(lib, hook, a_scale, w_scale):
    """the emulation splits x = x1 + x2 + x3 into bf16 planes: x3 = bf16(x - x1 - x2) sits 16 binades below x, so it
    leaves the bf16 normal range 16 binades before fp32 does.  Operands far from 1 in both directions (products kept
    inside the fp32 range): the error against fp64, relative to sum |a||b|, stays at the exact engine's level."""
    M, N, K = 600, 256, 1024
    g = torch.Generator(device="cuda").manual_seed(int(abs(np.log10(a_scale)) * 7 + abs(np.log10(w_scale))))
    A = (torch.randn(M + 1, K, device="cuda", generator=g) * a_scale).contiguous()
    W = (torch.randn(N, K, device="cuda", generator=g) * w_scale).contiguous()
    C3 = torch.full((M, N), float("nan"), device="cuda")
    C1 = torch.full((M, N), float("nan"), device="cuda")
    assert getattr(lib, hook)(_p(A), K, None, _p(W), K, None, None, _p(C3), M, N, K, 0, None) == 0
    assert lib.sttran_debug_gemm_padded(_p(A), K, None, _p(W), K, None, None, _p(C1), M, N, K, 0, 0, None) == 0
    torch.cuda.synchronize()
    ref = A[:M].double() @ W.double().T
    mag = (A[:M].double().abs() @ W.double().abs().T)            # sum |a||b| per output: what rounding errors scale with
    e3 = ((C3.double() - ref).abs() / mag).max().item()
    e1 = ((C1.double() - ref).abs() / mag).max().item()
    assert torch.isfinite(C3).all() and torch.isfinite(C1).all()
    assert e1 < 2e-6 and e3 < 2e-6, (e3, e1)                     # ~ sqrt(K) * 2^-24 for either engine
    assert e3 <= 4.0 * e1 + 1e-8, (e3, e1)


@pytest.mark.parametrize("hook", X3_HOOKS)
def test_gemm_bf16x3_cancellation(lib, hook):
    """mixed signs with heavy cancellation: every output is the difference of two nearly equal large sums, so an engine
    that lost low-order bits of the operands would show it.  a = [u | -u + d], w = [v | v]: a . w = d . v exactly in real
    arithmetic, |d| ~ 1e-4 |u|"""
    M, N, K = 520, 128, 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.randn(M + 1, K // 2, device="cuda", generator=g) * 100.0
    d = torch.randn(M + 1, K // 2, device="cuda", generator=g) * 0.01
    A = torch.cat([u, -u + d], dim=1).contiguous()
    v = torch.randn(N, K // 2, device="cuda", generator=g)
    W = torch.cat([v, v], dim=1).contiguous()
    C3 = torch.empty(M, N, device="cuda"); C1 = torch.empty(M, N, device="cuda")
    assert getattr(lib, hook)(_p(A), K, None, _p(W), K, None, None, _p(C3), M, N, K, 0, None) == 0
    assert lib.sttran_debug_gemm_padded(_p(A), K, None, _p(W), K, None, None, _p(C1), M, N, K, 0, 0, None) == 0
    torch.cuda.synchronize()
    ref = A[:M].double() @ W.double().T
    e3 = (C3.double() - ref).abs().max().item()
    e1 = (C1.double() - ref).abs().max().item()
    # both engines round the PRODUCTS' sum in fp32 (|u||v| K ~ 1e5 -> ~1e-2 absolute); neither may be worse than that
    assert e1 < 0.05 and e3 <= 2.0 * e1 + 1e-3, (e3, e1)


@pytest.mark.parametrize("hook", X3_HOOKS)
def test_gemm_bf16x3_subnormal_tail_is_documented(lib, hook):
    """where the emulation STOPS being fp32: operands below ~2^-110 have their third plane (and then the second) in the
    bf16 subnormal range, which the matrix pipe flushes -- the result degrades towards single-bf16 precision while the
    exact engine (whose products here are still normal fp32 numbers) keeps its accuracy.  Pinned so that a change of
    this behaviour is noticed; STTran activations are O(1) (LayerNorm outputs), ten binades of headroom are 30x more
    than they need."""
    M, N, K = 520, 128, 512
    g = torch.Generator(device="cuda").manual_seed(6)
    A = (torch.randn(M + 1, K, device="cuda", generator=g) * 1e-36).contiguous()
    W = (torch.randn(N, K, device="cuda", generator=g) * 1e30).contiguous()
    C3 = torch.empty(M, N, device="cuda"); C1 = torch.empty(M, N, device="cuda")
    assert getattr(lib, hook)(_p(A), K, None, _p(W), K, None, None, _p(C3), M, N, K, 0, None) == 0
    assert lib.sttran_debug_gemm_padded(_p(A), K, None, _p(W), K, None, None, _p(C1), M, N, K, 0, 0, None) == 0
    torch.cuda.synchronize()
    ref = A[:M].double() @ W.double().T
    mag = A[:M].double().abs() @ W.double().abs().T
    e3 = ((C3.double() - ref).abs() / mag).max().item()
    e1 = ((C1.double() - ref).abs() / mag).max().item()
    assert e1 < 2e-6                                           # the exact engine is unaffected
    assert e3 < 1e-2                                           # the emulation: still a bf16-grade answer, no NaN / Inf
    assert torch.isfinite(C3).all()


def test_gemm_bf16x3_t16_gathered_rows_and_determinism(lib):
    """the 16x16x32 kernel with a gathered A operand (the gather happens in the split pass), a row count that is no
    multiple of the 128-row tile or the 16-row block, stream-K ranges (176 columns x many K-steps) -- bit-identical on repeat"""
    M, N, K, R = 1000, 1936, 2048, 700
    g = torch.Generator(device="cuda").manual_seed(17)
    A = torch.randn(R, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    idx = torch.randint(0, R, (M,), device="cuda", generator=g, dtype=torch.int32)
    outs = []
    for _ in range(3):
        C3 = torch.full((M, N), float("nan"), device="cuda")
        assert lib.sttran_debug_gemm_emulated_t16(_p(A), K, _p(idx), _p(W), K, None, None, _p(C3), M, N, K, 0, None) == 0
        outs.append(C3)
    torch.cuda.synchronize()
    ref = A.double()[idx.long()] @ W.double().T
    assert (outs[0].double() - ref).abs().max().item() < 2e-3 and torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # a shape the tiles do not serve is refused, not mis-computed
    C = torch.empty(8, 100, device="cuda")
    assert lib.sttran_debug_gemm_emulated_t16(_p(A), K, None, _p(W), K, None, None, _p(C), 8, 100, K, 0, None) != 0


# ---- DSG-DETR on the device: class sequences (lib/dsg_detr.py:545-555) and attention over lengths the host never sees ---
def _dsg_layout_ref(pair_idx, labels, clip_start, NC):
    """numpy restatement in the reference's own terms: per clip, per class present, the pairs in pair order; position
    indices `[0]*count_0 + [1]*count_1 + ...` over the sorted unique subjects (lib/dsg_detr.py:551-554)"""
    P = pair_idx.shape[0]
    nclips = len(clip_start) - 1
    dec_off = np.zeros(nclips * NC, np.int32); dec_len = np.zeros(nclips * NC, np.int32)
    dec_src = np.zeros(P, np.int32); need = np.zeros(P, np.int32); out_src = np.zeros(P, np.int32)
    tok = 0
    for c in range(nclips):
        s, e = clip_start[c], clip_start[c + 1]
        cls = labels[pair_idx[s:e, 1]]
        for k in range(NC):
            idx = np.nonzero(cls == k)[0] + s
            dec_off[c * NC + k] = tok; dec_len[c * NC + k] = len(idx)
            if len(idx):
                _, cnt = np.unique(pair_idx[idx, 0], return_counts=True)
                dec_src[tok:tok + len(idx)] = idx
                need[tok:tok + len(idx)] = np.repeat(np.arange(len(cnt)), cnt)
                out_src[idx] = P + tok + np.arange(len(idx))
                tok += len(idx)
    return dec_off, dec_len, dec_src, need, out_src


@pytest.mark.parametrize("seed,clip_pairs,B", [(1, [7], 12), (2, [30, 1, 55], 40), (3, [200, 3], 64), (4, [1, 1, 1, 1], 5),
                                               (5, [480], 90), (6, [0, 9, 0, 4], 10),
                                               # many clips: large pair offsets with the per-clip tables in LDS (an LDS pointer
                                               # moved back by the clip's offset once faulted here), and a clip too big for LDS
                                               (7, [176] * 64, 500), (8, [7000, 50], 3000)])
def test_dsg_layout_on_device(lib, seed, clip_pairs, B):
    """random pair lists (subjects in ANY order, repeated subjects, classes 0..36, an empty clip) against the restatement"""
    rng = np.random.default_rng(seed)
    NC, P = 37, int(sum(clip_pairs))
    pair = np.stack([rng.integers(0, B, P), rng.integers(0, B, P)], axis=1).astype(np.int64)
    labels = rng.integers(0, NC, B).astype(np.int64)
    if seed == 5:
        labels[:] = 7                                                    # one 480-token sequence, few distinct subjects
    clip_start = np.concatenate([[0], np.cumsum(clip_pairs)]).astype(np.int32)
    K = (len(clip_pairs)) * NC
    dev = lambda a: torch.from_numpy(a).cuda()
    d_pair, d_lab, d_cs = dev(pair), dev(labels), dev(clip_start)
    outs = [torch.full((n,), -7, dtype=torch.int32, device="cuda") for n in (K, K, P, P, P)]
    scratch = torch.zeros(4 * P, dtype=torch.int32, device="cuda")
    err = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_dsg_layout(_p(d_pair), _p(d_lab), B, _p(d_cs), len(clip_pairs), NC, P, 400, *[_p(o) for o in outs],
                                       _p(scratch), _p(err), None) == 0
    torch.cuda.synchronize()
    ref = _dsg_layout_ref(pair, labels, clip_start, NC)
    for name, o, r in zip(("dec_off", "dec_len", "dec_src", "need", "out_src"), outs, ref):
        got = o.cpu().numpy()
        if name == "dec_off":                                            # the offset of an empty slot is arbitrary
            m = ref[1] > 0
            np.testing.assert_array_equal(got[m], r[m], err_msg=name)
        else:
            np.testing.assert_array_equal(got, r, err_msg=name)
    assert int(err[0]) == 0


def test_dsg_layout_flags_bad_indices_and_long_sequences(lib):
    NC, B = 37, 8
    pair = np.array([[0, 1], [2, 99], [3, 4]], np.int64)                 # an object box out of range
    labels = np.array([1, 5, 1, 1, 40, 1, 1, 1], np.int64)               # ... and a label out of range
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for n in (NC, NC, 3, 3, 3)]
    err = torch.zeros(16, dtype=torch.int32, device="cuda")
    cs = torch.tensor([0, 3], dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_dsg_layout(_p(torch.from_numpy(pair).cuda()), _p(torch.from_numpy(labels).cuda()), B, _p(cs), 1, NC, 3,
                                       400, *[_p(o) for o in outs], _p(torch.zeros(12, dtype=torch.int32, device="cuda")),
                                       _p(err), None) == 0
    torch.cuda.synchronize()
    assert int(err[0]) & 1
    # 5 distinct subjects in one class but only 3 position rows: clamped and flagged
    pair = np.stack([np.arange(5), np.full(5, 6)], axis=1).astype(np.int64)
    labels = np.full(8, 3, np.int64)
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for n in (NC, NC, 5, 5, 5)]
    err.zero_()
    cs = torch.tensor([0, 5], dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_dsg_layout(_p(torch.from_numpy(pair).cuda()), _p(torch.from_numpy(labels).cuda()), B, _p(cs), 1, NC, 5,
                                       3, *[_p(o) for o in outs], _p(torch.zeros(20, dtype=torch.int32, device="cuda")),
                                       _p(err), None) == 0
    torch.cuda.synchronize()
    assert int(err[0]) == 2 and outs[3].cpu().tolist() == [0, 1, 2, 2, 2]


@pytest.mark.parametrize("lens,bound", [([5, 0, 17, 48, 49, 0, 80, 3], 80), ([30, 81, 0, 200, 12, 64], 230), ([1, 2, 3], 40),
                                        ([0, 0, 0], 100), ([48, 16], 48),
                                        # round 5: enough slots to fill the chip (the <= 32 class on the high-residency form whatever
                                        # the bound), the two upper short classes in one launch, the general kernel walking its query
                                        # tiles in one workgroup per (head, slot)
                                        ([32, 5, 0, 17, 1, 0, 29, 8] * 40, 176), ([3, 0, 40, 0, 75, 0, 0, 6] * 35, 176),
                                        ([0, 100, 2, 0, 161, 33] * 45, 176)])
def test_attention_over_device_side_lengths(lib, lens, bound):
    """every length class in one call, empty slots included; rows of no sequence stay untouched"""
    dim, nhead = 1936, 8
    g = torch.Generator(device="cuda").manual_seed(sum(lens) + bound)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    tokens = max(int(sum(lens)), 1)
    qkv = torch.randn(tokens, 3 * dim, device="cuda", generator=g)
    out = torch.full((tokens, dim), float("nan"), device="cuda")
    so = torch.from_numpy(offs).cuda()
    sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    assert lib.sttran_debug_attention_classes(_p(qkv), _p(so), _p(sl), len(lens), bound, _p(out), dim, nhead, None) == 0
    torch.cuda.synchronize()
    if sum(lens) == 0:
        assert torch.isnan(out).all()
        return
    keep = [i for i, l in enumerate(lens) if l > 0]
    ref = _attn_ref(qkv, [int(offs[i]) for i in keep], [lens[i] for i in keep], dim, nhead)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 2e-5


def test_gemm_t16_operand_larger_than_4_gb(lib):
    """gemm16_kernel stages through 32-bit byte offsets.  They are taken from a 64-bit base per TILE now, so an
    activation operand of more than 4 GB (550 k rows at the workspace's row stride of 1952 floats: decoder tokens of
    ~275 k pairs, which 288 GB hold) is read where it is -- and a GATHERED operand, whose offsets are taken from the
    operand's base, only reaches the kernel when its span is known to stay below 4 GB (else the general engine, which
    carries 64-bit pointers, serves it).  Both cases: rows on either side of the 4 GB mark against an fp64 reference."""
    K, N, ld = 1936, 176, 1952
    M = (1 << 32) // (ld * 4) + 3000                   # ~553 k rows: the operand ends ~23 MB past 4 GB
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.zeros(M, ld, device="cuda")
    probe = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 3000, M)]).cuda()
    A[probe, :K] = torch.randn(len(probe), K, device="cuda", generator=g)
    W = torch.zeros(N, ld, device="cuda")
    W[:, :K] = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    Cc = torch.full((M, N), float("nan"), device="cuda")
    assert lib.sttran_debug_plan_tile(M, N, K) == 5
    assert lib.sttran_debug_gemm_padded(_p(A), ld, None, _p(W), ld, _p(b), None, _p(Cc), M, N, K, 0, 5, None) == 0
    torch.cuda.synchronize()
    ref = A[probe, :K].double() @ W[:, :K].double().T + b.double()
    assert (Cc[probe].double() - ref).abs().max().item() < 2e-4
    assert torch.isfinite(Cc).all() and (Cc[1000] == b).all()              # an all-zero row: exactly the bias
    # gathered rows from beyond the 4 GB mark (tile 5 requested: the dispatcher must not wrap)
    idx = probe.flip(0).to(torch.int32).contiguous()
    C2 = torch.full((len(idx), N), float("nan"), device="cuda")
    assert lib.sttran_debug_gemm_padded(_p(A), ld, _p(idx), _p(W), ld, _p(b), None, _p(C2), len(idx), N, K, 0, 5, None) == 0
    torch.cuda.synchronize()
    assert (C2.double() - ref.flip(0)).abs().max().item() < 2e-4
    # ... and gathered rows of a small operand still run on the 16x16x4 tile, bit-identical to the ungathered launch
    small = A[probe].contiguous()
    ident = torch.arange(len(probe), device="cuda", dtype=torch.int32)
    C3, C4 = torch.empty(len(probe), N, device="cuda"), torch.empty(len(probe), N, device="cuda")
    assert lib.sttran_debug_gemm_padded(_p(small), ld, None, _p(W), ld, _p(b), None, _p(C3), len(probe), N, K, 0, 5, None) == 0
    assert lib.sttran_debug_gemm_padded(_p(small), ld, _p(ident), _p(W), ld, _p(b), None, _p(C4), len(probe), N, K, 0, 5, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(C3, C4)                         # same launch geometry: the gathered read changes nothing
    assert (C3 - Cc[probe]).abs().max().item() < 1e-4  # (another M = another stream-K cut of the K sum: close, not identical)


# ---- mask_conv1_pool_kernel: Conv2d(2,128,7,s2,p3) -> ReLU -> BN(eval) -> MaxPool2d(3,2,1), one kernel (lib/sttran.py:337-341) ----
def _pack_w0(w0, b0):
    """conv.0.weight [128][2][7][7] -> [128][13 groups][2 channels][4 taps]; tap 49 = (bias, 0) (csrc/api_weights.hip)"""
    w = w0.reshape(128, 2, 49).cpu().numpy()
    wp = np.zeros((128, 13, 2, 4), np.float32)
    for t in range(49):
        wp[:, t // 4, :, t % 4] = w[:, :, t]
    wp[:, 49 // 4, 0, 49 % 4] = b0.cpu().numpy()
    return torch.from_numpy(wp.reshape(128, 104)).cuda()


@pytest.mark.parametrize("P", [1, 2, 7, 176, 515, 1031])
def test_mask_conv1_pool_against_torch_fp64(lib, P):
    """every output position of every channel (the kernel's column -> position map, parity sub-plane layout of the padded
    masks, pooling windows incl. the -inf border) against F.conv2d / batch_norm / max_pool2d in float64, incl. pair counts
    that are no multiple of the grid and masks scattered through per-pair offsets (the by-pointer batch form)"""
    F = torch.nn.functional
    g = torch.Generator(device="cuda").manual_seed(900 + P)
    masks = torch.rand(P, 2, 27, 27, device="cuda", generator=g) - 0.5
    w0 = (torch.rand(128, 2, 7, 7, device="cuda", generator=g) - 0.5) * 0.2
    b0 = torch.rand(128, device="cuda", generator=g) - 0.5
    scale = torch.rand(128, device="cuda", generator=g) + 0.5
    scale[::3] *= -1                                         # a negative BN gain reverses the order inside a pooling window
    shift = torch.rand(128, device="cuda", generator=g) - 0.5
    ref = F.conv2d(masks.double(), w0.double(), b0.double(), stride=2, padding=3).clamp_min(0)
    ref = F.max_pool2d(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], 3, 2, 1)
    ref = ref.permute(0, 2, 3, 1).contiguous()              # channel-last [P,7,7,128]
    wp = _pack_w0(w0, b0)
    out = torch.full((P, 7, 7, 128), float("nan"), device="cuda")
    assert lib.sttran_debug_mask_conv1_pool(_p(masks), None, _p(wp), _p(scale), _p(shift), _p(out), P, None) == 0
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() < 2e-5
    # the same pairs stored in reverse order with gaps, addressed through mask_off
    pool = torch.zeros(P * 1500 + 64, device="cuda")
    off = torch.tensor([(P - 1 - p) * 1500 + 8 for p in range(P)], device="cuda", dtype=torch.int64)
    for p in range(P):
        pool[int(off[p]):int(off[p]) + 1458] = masks[p].reshape(-1)
    out2 = torch.full((P, 7, 7, 128), float("nan"), device="cuda")
    assert lib.sttran_debug_mask_conv1_pool(_p(pool), _p(off), _p(wp), _p(scale), _p(shift), _p(out2), P, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_mask_conv1_pool_propagates_nan_like_torch(lib):
    """torch's max_pool2d returns NaN for a window that holds one, ReLU keeps NaN: one poisoned mask element must poison
    exactly the pooled outputs whose receptive field contains it, in every channel"""
    F = torch.nn.functional
    g = torch.Generator(device="cuda").manual_seed(77)
    masks = torch.rand(3, 2, 27, 27, device="cuda", generator=g) - 0.5
    masks[1, 1, 13, 5] = float("nan")
    w0 = (torch.rand(128, 2, 7, 7, device="cuda", generator=g) - 0.5) * 0.2
    b0 = torch.rand(128, device="cuda", generator=g) - 0.5
    scale = torch.rand(128, device="cuda", generator=g) + 0.5
    shift = torch.rand(128, device="cuda", generator=g) - 0.5
    # the reference's semantics are torch-CPU's (a GPU library's pooling need not propagate NaN): computed on the host
    mc, wc, bc, sc, tc = (t.cpu() for t in (masks, w0, b0, scale, shift))
    ref = F.conv2d(mc, wc, bc, stride=2, padding=3)
    ref = torch.where(torch.isnan(ref), ref, ref.clamp_min(0))
    ref = F.max_pool2d(ref * sc[None, :, None, None] + tc[None, :, None, None], 3, 2, 1).permute(0, 2, 3, 1).cuda()
    out = torch.zeros(3, 7, 7, 128, device="cuda")
    assert lib.sttran_debug_mask_conv1_pool(_p(masks), None, _p(_pack_w0(w0, b0)), _p(scale), _p(shift), _p(out), 3, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.isnan(out).any() and not torch.isnan(out[0]).any()
    ok = ~torch.isnan(ref)
    assert (out[ok] - ref[ok]).abs().max().item() < 2e-5
