"""Parity of every hot-path command with the reference's own CPU backend (oracle/_ref/libccv_ref.so) on identical
seeded inputs.  Runs on the CPU HIP emulator in the `not gpu` tier and on the MI355X in the `gpu` tier.
Tolerances: fp32 conv/GEMM 1e-4 relative (north_star); pooling, relu, transfers bit-exact."""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_pair, exec_on, out_hw, tensor_eq, make_tensors

F = np.float32


def rnd(rng, *shape, scale=1.0):
    return (rng.random(shape, dtype=F) * scale).astype(F)


def srnd(rng, *shape, scale=1.0):
    return ((rng.random(shape, dtype=F) - 0.5) * 2 * scale).astype(F)


CONV_CASES = [
    # n, h, w, c, k, kh, kw, stride, border, groups, dilation, bias
    (2, 9, 10, 8, 16, 3, 3, (1, 1), (1, 1), 1, None, True),      # vectorised gather, ragged M
    (1, 12, 11, 3, 5, 5, 5, (2, 2), (2, 2), 1, None, True),      # C=3 scalar gather, stride 2, ragged K
    (2, 7, 7, 12, 8, 1, 1, (1, 1), (0, 0), 1, None, False),      # 1x1, no bias
    (8, 7, 7, 256, 264, 1, 1, (1, 1), (0, 0), 1, None, True),    # 1x1 as two plain matrices (conv_pointwise), forward on the bf16 pipe's exact split (K >= 128, 256 x 256 outputs)
    (1, 15, 13, 4, 6, 7, 7, (2, 2), (3, 3), 1, None, True),      # 7x7 s2 (ResNet stem shape class)
    (2, 8, 8, 8, 8, 3, 3, (1, 1), (1, 1), 2, None, True),        # groups
    (1, 11, 11, 4, 4, 3, 3, (1, 1), (2, 2), 1, (2, 2), True),    # dilation
    (3, 6, 20, 40, 130, 3, 3, (1, 1), (1, 1), 1, None, True),    # several K-steps, two N tiles, >1 M tile
    (2, 12, 13, 32, 64, 3, 3, (2, 2), (1, 1), 1, None, True),    # stride 2 with >= 32 channels: the division-free (incremental) strided dgrad
    (2, 9, 9, 64, 32, 3, 3, (1, 1), (2, 2), 1, (2, 2), False),   # dilation 2 on the incremental path
    (5, 4, 3, 32, 32, 3, 3, (1, 1), (1, 1), 1, None, True),      # tiny maps: several image wraps per K-step (incremental wgrad must fall back)
]


def _conv_inputs(case, seed=0):
    n, h, w, c, k, kh, kw, stride, border, groups, dil, bias = case
    rng = np.random.default_rng(seed)
    a = srnd(rng, n, h, w, c)
    wt = srnd(rng, k, kh, kw, c // groups, scale=1.0 / (kh * kw * (c // groups)))
    b = srnd(rng, k) if bias else None
    hint = nnc.HINT(stride, border)
    ekh, ekw = ((kh - 1) * dil[0] + 1, (kw - 1) * dil[1] + 1) if dil else (kh, kw)
    oh, ow = out_hw(h, w, ekh, ekw, hint)
    return a, wt, b, hint, oh, ow


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(backend, ref_lib, case):
    n, h, w, c, k, kh, kw, stride, border, groups, dil, bias = case
    a, wt, b, hint, oh, ow = _conv_inputs(case)
    cmd = nnc.CMD_CONVOLUTION_FORWARD(groups, k, kh, kw, c // groups, dilation=dil)
    ins = [a, wt] + ([b] if bias else [])
    got, want = exec_pair(backend, ref_lib, cmd, hint, 0, ins, [np.zeros((n, oh, ow, k), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("flags", [0, nnc.ACCUMULATE_OUTPUT])
def test_conv_backward(backend, ref_lib, case, flags):
    n, h, w, c, k, kh, kw, stride, border, groups, dil, bias = case
    a, wt, b, hint, oh, ow = _conv_inputs(case)
    rng = np.random.default_rng(7)
    g = srnd(rng, n, oh, ow, k)
    g[g < -0.6] = 0  # exercise the oracle's v == 0 shortcut
    dw0 = srnd(rng, *wt.shape)
    db0 = srnd(rng, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(groups, k, kh, kw, c // groups, dilation=dil)
    got, want = exec_pair(backend, ref_lib, cmd, hint, flags, [g, a, wt], [np.zeros_like(a), dw0, db0])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5)
    if flags & nnc.ACCUMULATE_OUTPUT:
        # Quirk: the CPU oracle OVERWRITES dbias even under CCV_NNC_ACCUMULATE_OUTPUT (conv_cpu_ref.c:262-263, `bias[k] = biasval`),
        # while the GPU backend being replaced accumulates it (conv_gpu_cudnn.cu:264-271, beta = 1).  We follow the GPU
        # backend: expected = initial + the oracle's plain bias gradient.
        np.testing.assert_allclose(got[2], db0 + want[2], rtol=1e-4, atol=2e-5)
    else:
        np.testing.assert_allclose(got[2], want[2], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("tile", [(2, 2), (2, 1), (1, 2), (1, 1)])
@pytest.mark.parametrize("idx", [1, 6, 7])
def test_conv_every_block_tile(backend, ref_lib, tile, idx):
    """The launcher picks the block tile from the problem size; force each shape over forward, dgrad and wgrad (incl.
    split-K) so every (loader, tile) instantiation is checked, not only the one the size heuristic lands on."""
    case = CONV_CASES[idx]
    n, h, w, c, k, kh, kw, stride, border, groups, dil, bias = case
    a, wt, b, hint, oh, ow = _conv_inputs(case)
    g = srnd(np.random.default_rng(11), n, oh, ow, k)
    backend.force_tile(*tile)
    try:
        fwd = nnc.CMD_CONVOLUTION_FORWARD(groups, k, kh, kw, c // groups, dilation=dil)
        got, want = exec_pair(backend, ref_lib, fwd, hint, 0, [a, wt, b], [np.zeros((n, oh, ow, k), F)])
        np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
        bwd = nnc.CMD_CONVOLUTION_BACKWARD(groups, k, kh, kw, c // groups, dilation=dil)
        got, want = exec_pair(backend, ref_lib, bwd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)])
        for i in range(3):
            np.testing.assert_allclose(got[i], want[i], rtol=1e-4, atol=2e-5)
    finally:
        backend.force_tile(0, 0)


WINO_CASES = [
    # n, h, w, c, k, border  (3x3, stride 1): F(4x4, 3x3) tiles incl. ragged edge tiles and every legal padding
    (2, 8, 8, 8, 16, (1, 1)),      # exact 2x2 tiles
    (1, 13, 13, 32, 32, (1, 1)),   # VGG conv5 geometry: 13 -> 4 tiles of 4 with 3 clipped rows / columns
    (3, 9, 14, 12, 20, (0, 0)),    # no padding: output 7 x 12
    (2, 5, 7, 8, 8, (2, 2)),       # full padding: output 7 x 9
    (1, 27, 27, 64, 48, (1, 1)),   # several GEMM K-steps, more than one 128-row tile of tiles
    (2, 6, 6, 4, 4, (1, 0)),       # asymmetric begin padding
]


def _wino_pair(backend, ref_lib, cmd, hint, ins, outs):
    """ours under algorithm 1 (Winograd), the reference's CPU_REF direct loops under its default"""
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, ins, outs, backend=nnc.BACKEND_CPU_REF)
    cmd.algorithm = 1
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, ins, outs)
    cmd.algorithm = -1
    assert r1 == 0 and r2 == 0, (r1, r2)
    return got, want


def _wino_inputs(case, seed=0):
    n, h, w, c, k, border = case
    rng = np.random.default_rng(seed)
    a = srnd(rng, n, h, w, c)
    wt = srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c))
    b = srnd(rng, k)
    hint = nnc.HINT((1, 1), border)
    oh, ow = out_hw(h, w, 3, 3, hint)
    return a, wt, b, hint, oh, ow


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_forward_winograd(backend, ref_lib, case):
    """cmd.algorithm = 1: the Winograd F(4x4, 3x3) path against the reference's direct convolution (1e-4 relative: the
    reference holds its own CPU Winograd to the same REQUIRE_TENSOR_EQ, test/unit/nnc/winograd.tests.c:37-130)."""
    n, h, w, c, k, border = case
    a, wt, b, hint, oh, ow = _wino_inputs(case)
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    got, want = _wino_pair(backend, ref_lib, cmd, hint, [a, wt, b], [np.zeros((n, oh, ow, k), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
    got, want = _wino_pair(backend, ref_lib, cmd, hint, [a, wt], [np.full((n, oh, ow, k), 7, F)])  # no bias; stale output must be overwritten
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_backward_winograd_dgrad(backend, ref_lib, case):
    """cmd.algorithm = 1 on the backward row: the data gradient through Winograd F(4x4,3x3) (mirrored taps, padding 2 - p),
    the filter gradient through F(3x3,4x4) with the batched split-K contraction over tiles, against the reference."""
    n, h, w, c, k, border = case
    a, wt, b, hint, oh, ow = _wino_inputs(case)
    g = srnd(np.random.default_rng(5), n, oh, ow, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    got, want = _wino_pair(backend, ref_lib, cmd, hint, [g, a, wt], [np.full_like(a, 3), np.zeros_like(wt), np.zeros(k, F)])
    # 1e-4 relative; the absolute floor scales with the tensor: the Winograd transforms add and subtract terms weighted up
    # to 8x before the products, so the error follows the magnitude of the TERMS, not of a sum that cancelled to ~0 (the
    # signed random gradients here cancel heavily in dw; observed worst case 1.1e-5 of the tensor's scale)
    tol = lambda ref: dict(rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    for i in range(3):
        np.testing.assert_allclose(got[i], want[i], **tol(want[i]))
    # CCV_NNC_ACCUMULATE_OUTPUT: dw accumulates on top of its previous contents
    dw0 = srnd(np.random.default_rng(6), *wt.shape)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, nnc.ACCUMULATE_OUTPUT, [g, a, wt], [np.zeros_like(a), dw0.copy(), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    cmd.algorithm = 1
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, nnc.ACCUMULATE_OUTPUT, [g, a, wt], [np.zeros_like(a), dw0.copy(), np.zeros(k, F)])
    assert r1 == 0 and r2 == 0
    np.testing.assert_allclose(got[1], want[1], **tol(want[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 55, 55, 256, 256), (4, 111, 111, 64, 128), (16, 13, 13, 512, 512), (64, 111, 111, 64, 64)])
def test_conv_winograd_matches_implicit_gemm_at_vgg_sizes(gpu_lib, shape):
    """At VGG-D layer sizes (beyond what the CPU oracle finishes): the two algorithms of the conv rows agree to 1e-4 of the
    output scale, forward and both gradients -- the implicit GEMM being the path the oracle pins at small sizes.  The
    last shape has 803 K transform threads: beyond the 2048-block cap of the grid-stride launch helper (a transform launched
    through it silently covered only the first 524 K -- caught by the PMC write counter, not by the smaller shapes)."""
    n, h, w, c, k = shape
    rng = np.random.default_rng(2)
    a, wt, b = srnd(rng, n, h, w, c), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c) ** 0.5), srnd(rng, k)
    g = srnd(rng, n, h, w, k)
    hint = nnc.HINT((1, 1), (1, 1))
    res = {}
    for algo in (0, 1):
        f = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
        f.algorithm = algo
        r, out = exec_on(gpu_lib, nnc.GPU_MEMORY, f, hint, 0, [a, wt, b], [np.zeros((n, h, w, k), F)])
        assert r == 0
        bw = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
        bw.algorithm = algo
        r, outb = exec_on(gpu_lib, nnc.GPU_MEMORY, bw, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)])
        assert r == 0
        res[algo] = (out[0], outb[0], outb[1])
    for i in range(3):
        scale = float(np.abs(res[0][i]).max())
        assert float(np.abs(res[0][i] - res[1][i]).max()) <= 1e-4 * scale, (i, scale)


# CONVOLUTION_TRANSPOSE_FORWARD (a registered row without a test).  a [n][oh][ow][Ca] is scattered through w [Ca][kh][kw][count / groups]
# into b [n][H][W][count], where (H, W) -> (oh, ow) is the convolution the hint describes (convolution/ccv_nnc_conv_transpose_cpu_ref.c:13-; the
# reference's GPU cases: test/int/nnc/cudnn.tests.c:5338-5500, 7 x 7 stride 2 on 224 x 224, tolerance 2e-4 there).
#               n, H,  W,  count, Ca, k, stride, border, groups, dilation, bias, fmt
CONVT_CASES = [(2, 9,  9,  16,    8,  3, 1,      1,      1,      1,        True,  "NHWC"),
               (2, 13, 13, 3,     6,  7, 2,      3,      1,      1,        True,  "NHWC"),   # the reference case's geometry, small
               (3, 8,  10, 8,     16, 2, 2,      0,      2,      1,        False, "NHWC"),   # the up-sampling deconvolution, two groups
               (2, 9,  11, 12,    8,  3, 2,      1,      1,      1,        True,  "NCHW"),
               (1, 12, 12, 8,     4,  3, 1,      2,      1,      2,        True,  "NHWC")]   # dilated taps


@pytest.mark.parametrize("case", CONVT_CASES, ids=[str(c) for c in CONVT_CASES])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_conv_transpose_forward(backend, ref_lib, case, half):
    n, H, W, count, ca, k, stride, border, groups, dil, with_bias, fmt = case
    rng = np.random.default_rng(7)
    ke = (k - 1) * dil + 1
    oh, ow = (H + 2 * border - ke) // stride + 1, (W + 2 * border - ke) // stride + 1
    cg = count // groups
    a = srnd(rng, n, oh, ow, ca)
    w = srnd(rng, ca, k, k, cg, scale=1.0 / (k * k * ca))
    bias = srnd(rng, count, scale=0.5) if with_bias else None
    out = np.zeros((n, H, W, count), F)
    if fmt == "NCHW":
        a, w, out = a.transpose(0, 3, 1, 2).copy(), w.transpose(0, 3, 1, 2).copy(), out.transpose(0, 3, 1, 2).copy()
    cmd = nnc.CMD_CONVOLUTION_TRANSPOSE_FORWARD(groups, count, 0, k, k, ca, dilation=(dil, dil) if dil > 1 else None)
    hint = nnc.HINT((stride, stride), (border, border))
    if half:  # the oracle runs in fp32 on the half-rounded inputs (tests/test_half.py's rule); bound 5e-3 as the reference's half case (cudnn.tests.c:5500)
        a, w = a.astype(np.float16), w.astype(np.float16)
        bias = bias.astype(np.float16) if bias is not None else None
        r1, (got,) = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, [a, w, bias], [out.astype(np.float16)], fmt)
        r2, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a.astype(F), w.astype(F), None if bias is None else bias.astype(F)], [out], fmt, backend=nnc.BACKEND_CPU_REF)
        assert r1 == 0 and r2 == 0 and got.dtype == np.float16
        np.testing.assert_allclose(got.astype(F), want, rtol=0, atol=5e-3 * max(1.0, float(np.abs(want).max())))
        return
    got, want = exec_pair(backend, ref_lib, cmd, hint, 0, [a, w, bias], [out], fmt)
    assert np.abs(want).max() > 0.01
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_conv_backward_partial_outputs(backend, ref_lib):
    case = CONV_CASES[0]
    n, h, w, c, k, kh, kw, stride, border, groups, dil, bias = case
    a, wt, b, hint, oh, ow = _conv_inputs(case)
    g = srnd(np.random.default_rng(3), n, oh, ow, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(groups, k, kh, kw, c)
    got, want = exec_pair(backend, ref_lib, cmd, hint, 0, [g, a, wt], [None, np.zeros_like(wt), np.zeros(k, F)])  # no dgrad (conv1_1)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got[2], want[2], rtol=1e-4, atol=2e-5)
    got, want = exec_pair(backend, ref_lib, cmd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt)])  # no bias grad
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5)


def test_conv_forward_3d_no_batch(backend, ref_lib):
    rng = np.random.default_rng(1)
    a = srnd(rng, 10, 9, 4)
    wt = srnd(rng, 8, 3, 3, 4, scale=1 / 36)
    b = srnd(rng, 8)
    hint = nnc.HINT((1, 1), (1, 1))
    got, want = exec_pair(backend, ref_lib, nnc.CMD_CONVOLUTION_FORWARD(1, 8, 3, 3, 4), hint, 0, [a, wt, b], [np.zeros((10, 9, 8), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)


GEMM_CASES = [
    # (a shape, w shape, transpose_a, transpose_b, bias)
    ((5, 3), (3, 7), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, True),
    ((6, 20), (9, 20), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True),      # cnnp dense layer: w [out][in]
    ((20, 6), (20, 9), nnc.TRANSPOSE(0, 1), nnc.NO_TRANSPOSE, False),
    ((12, 4), (8, 12), nnc.TRANSPOSE(0, 1), nnc.TRANSPOSE(0, 1), True),
    ((130, 100), (257, 100), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True),  # several tiles, ragged
    ((8, 2048), (16, 2048), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True),   # split-K path
    ((3, 4, 5), (3, 5, 6), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, False),      # batched
    ((3, 4, 5), (6, 5), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True),       # batched a, shared w
]


def _gemm_shapes(ashape, wshape, ta, tb):
    ar, ac = ashape[-2:]
    if ta[0] != ta[1]:
        ar, ac = ac, ar
    wr, wc = wshape[-2:]
    if tb[0] != tb[1]:
        wr, wc = wc, wr
    assert ac == wr
    batch = ashape[:-2] if len(ashape) > 2 else (wshape[:-2] if len(wshape) > 2 else ())
    return batch + (ar, wc)


def _t(t, nd):
    """transpose axes are absolute indices: (0,1) for 2-d, (1,2) for 3-d."""
    return t if t[0] == t[1] or nd == 2 else (nd - 2, nd - 1)


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_forward(backend, ref_lib, case):
    ashape, wshape, ta, tb, bias = case
    rng = np.random.default_rng(0)
    a, w = srnd(rng, *ashape), srnd(rng, *wshape)
    ta, tb = _t(ta, len(ashape)), _t(tb, len(wshape))
    bshape = _gemm_shapes(ashape, wshape, ta, tb)
    ins = [a, w] + ([srnd(rng, bshape[-1])] if bias else [])
    got, want = exec_pair(backend, ref_lib, nnc.CMD_GEMM_FORWARD(ta, tb), nnc.NO_HINT, 0, ins, [np.zeros(bshape, F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", GEMM_CASES)
@pytest.mark.parametrize("flags", [0, nnc.ACCUMULATE_OUTPUT])
def test_gemm_backward(backend, ref_lib, case, flags):
    ashape, wshape, ta, tb, bias = case
    rng = np.random.default_rng(1)
    a, w = srnd(rng, *ashape), srnd(rng, *wshape)
    ta, tb = _t(ta, len(ashape)), _t(tb, len(wshape))
    bshape = _gemm_shapes(ashape, wshape, ta, tb)
    g = srnd(rng, *bshape)
    outs = [srnd(rng, *ashape), srnd(rng, *wshape)] + ([srnd(rng, bshape[-1])] if bias else [])
    got, want = exec_pair(backend, ref_lib, nnc.CMD_GEMM_BACKWARD(ta, tb), nnc.NO_HINT, flags, [g, a, w], outs)
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-4)


POOL_CASES = [
    # h, w, c, kh, kw, stride, border
    (9, 9, 5, 3, 3, (2, 2), (0, 0)),     # VGG-D pool
    (10, 12, 7, 3, 3, (2, 2), (1, 1)),   # ResNet stem pool
    (8, 8, 4, 2, 2, (2, 2), (0, 0)),
    (7, 7, 6, 7, 7, (1, 1), (0, 0)),     # global
    (6, 5, 3, 3, 3, (1, 1), (1, 1)),     # overlapping, stride 1
    (8, 6, 5, 2, 2, (2, 2), (0, 0)),     # windows that tile the map (the gradient's window-per-thread kernel), channels not in 16-byte groups
    (6, 9, 8, 2, 3, (2, 3), (0, 0)),     # tiling, 2 x 3 windows
    (6, 6, 4, 3, 3, (3, 3), (0, 0)),     # tiling, averages divided by 9
    (9, 7, 8, 2, 2, (2, 2), (0, 0)),     # tiling with a left-over row and column (VGG-D's 225 x 225 maps): zero gradient there
    (7, 8, 3, 3, 3, (3, 3), (0, 0)),     # left-over row of one, left-over columns of two
]


@pytest.mark.parametrize("case", POOL_CASES)
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_pool_forward_backward_bit_exact(backend, ref_lib, case, kind):
    h, w, c, kh, kw, stride, border = case
    rng = np.random.default_rng(2)
    a = np.round(srnd(rng, h, w, c) * 4) / 4  # ties on purpose: the max-pool gradient goes to every tied position
    hint = nnc.HINT(stride, border)
    oh, ow = out_hw(h, w, kh, kw, hint)
    fcmd = nnc.CMD_MAX_POOL_FORWARD(kh, kw) if kind == "max" else nnc.CMD_AVERAGE_POOL_FORWARD(kh, kw)
    bcmd = nnc.CMD_MAX_POOL_BACKWARD(kh, kw) if kind == "max" else nnc.CMD_AVERAGE_POOL_BACKWARD(kh, kw)
    got, want = exec_pair(backend, ref_lib, fcmd, hint, 0, [a], [np.zeros((oh, ow, c), F)])
    assert np.array_equal(got[0].view(np.int32), want[0].view(np.int32))
    g = srnd(rng, oh, ow, c)
    got, want = exec_pair(backend, ref_lib, bcmd, hint, 0, [g, a, want[0]], [np.zeros_like(a)])
    assert np.array_equal(got[0].view(np.int32), want[0].view(np.int32))


@pytest.mark.parametrize("geom", [(3, 9, 8, 5, 3, 3, (2, 2), (1, 1)), (3, 8, 10, 8, 2, 2, (2, 2), (0, 0)), (2, 9, 11, 4, 2, 2, (2, 2), (0, 0)),
                                  (2, 16, 16, 3, 3, 3, (2, 2), (1, 1)), (2, 8, 16, 5, 2, 2, (2, 2), (0, 0)), (2, 6, 12, 4, 2, 2, (2, 2), (0, 0)), (2, 7, 8, 3, 3, 3, (1, 1), (1, 1)), (2, 8, 8, 2, 8, 8, (1, 1), (0, 0))],
                         ids=["overlapping", "tiling", "tiling-leftover", "stem-3x3-s2-rows", "tiling-rows", "tiling-rows-odd-ow", "3x3-s1-rows", "global-rows"])
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_pool_batched_and_nchw(backend, ref_lib, kind, geom):
    """The CPU oracle walks only image 0 of a batch and is NHWC-only: check a batch image by image, and NCHW against
    the transposed NHWC result."""
    rng = np.random.default_rng(3)
    n, h, w, c, kh, kw, stride, border = geom
    hint = nnc.HINT(stride, border)
    oh, ow = out_hw(h, w, kh, kw, hint)
    a = np.round(srnd(rng, n, h, w, c) * 4) / 4
    g = srnd(rng, n, oh, ow, c)
    fcmd = nnc.CMD_MAX_POOL_FORWARD(kh, kw) if kind == "max" else nnc.CMD_AVERAGE_POOL_FORWARD(kh, kw)
    bcmd = nnc.CMD_MAX_POOL_BACKWARD(kh, kw) if kind == "max" else nnc.CMD_AVERAGE_POOL_BACKWARD(kh, kw)
    r, (b,) = exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [a], [np.zeros((n, oh, ow, c), F)])
    assert r == 0
    r, (hg,) = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, [g, a, b], [np.zeros_like(a)])
    assert r == 0
    for i in range(n):
        _, (bw,) = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, hint, 0, [a[i]], [np.zeros((oh, ow, c), F)], backend=nnc.BACKEND_CPU_REF)
        _, (hw,) = exec_on(ref_lib, nnc.CPU_MEMORY, bcmd, hint, 0, [g[i], a[i], bw], [np.zeros((h, w, c), F)], backend=nnc.BACKEND_CPU_REF)
        assert np.array_equal(b[i].view(np.int32), bw.view(np.int32))
        assert np.array_equal(hg[i].view(np.int32), hw.view(np.int32))
    ac, gc = np.ascontiguousarray(a.transpose(0, 3, 1, 2)), np.ascontiguousarray(g.transpose(0, 3, 1, 2))
    r, (bc,) = exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [ac], [np.zeros((n, c, oh, ow), F)], fmt="NCHW")
    assert r == 0 and np.array_equal(bc.transpose(0, 2, 3, 1), b)
    r, (hc,) = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, [gc, ac, bc], [np.zeros_like(ac)], fmt="NCHW")
    assert r == 0 and np.array_equal(hc.transpose(0, 2, 3, 1), hg)
    # the trainers' half-precision tensors: NCHW (rows of four per lane where the map allows) == NHWC (one lane per element), bit for bit
    H = np.float16
    a16, g16 = a.astype(H), g.astype(H)
    r, (b16,) = exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [a16], [np.zeros((n, oh, ow, c), H)])
    assert r == 0
    r, (h16,) = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, [g16, a16, b16], [np.zeros_like(a16)])
    assert r == 0
    ac16, gc16 = np.ascontiguousarray(a16.transpose(0, 3, 1, 2)), np.ascontiguousarray(g16.transpose(0, 3, 1, 2))
    r, (bc16,) = exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [ac16], [np.zeros((n, c, oh, ow), H)], fmt="NCHW")
    assert r == 0 and np.array_equal(bc16.transpose(0, 2, 3, 1), b16)
    r, (hc16,) = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, [gc16, ac16, bc16], [np.zeros_like(ac16)], fmt="NCHW")
    assert r == 0 and np.array_equal(hc16.transpose(0, 2, 3, 1), h16)


@pytest.mark.parametrize("shape", [(1,), (7,), (4, 5, 6, 3), (2, 1027)])
def test_relu(backend, ref_lib, shape):
    rng = np.random.default_rng(4)
    a = srnd(rng, *shape)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_RELU_FORWARD(), nnc.NO_HINT, 0, [a], [np.zeros(shape, F)])
    assert np.array_equal(got[0], want[0])
    g = srnd(rng, *shape)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_RELU_BACKWARD(), nnc.NO_HINT, 0, [g, None, want[0]], [np.zeros(shape, F)])
    assert np.array_equal(got[0], want[0])


@pytest.mark.parametrize("count", [1, 2, 3, 4, 5])
def test_ewsum(backend, ref_lib, count):
    rng = np.random.default_rng(5)
    ins = [srnd(rng, 3, 5, 7) for _ in range(count)]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_EWSUM_FORWARD(), nnc.NO_HINT, 0, ins, [np.zeros((3, 5, 7), F)])
    assert np.array_equal(got[0], want[0])
    g = srnd(rng, 3, 5, 7)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_EWSUM_BACKWARD(), nnc.NO_HINT, 0, [g] + ins + [want[0]], [np.zeros((3, 5, 7), F) for _ in range(count)])
    for x, y in zip(got, want):
        assert np.array_equal(x, y)


def test_scalar_mul_and_set(backend, ref_lib):
    rng = np.random.default_rng(6)
    a = srnd(rng, 33, 5)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SCALAR_MUL_FORWARD(0.3), nnc.NO_HINT, 0, [a], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0])
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SET_FORWARD(1.5), nnc.NO_HINT, 0, [], [srnd(rng, 9, 3)])
    assert np.array_equal(got[0], want[0])
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SET_FORWARD(0), nnc.NO_HINT, 0, [], [srnd(rng, 9, 3)])
    assert np.array_equal(got[0], want[0])


@pytest.mark.parametrize("nesterov,damp", [(0, 0.9), (0, 0.0), (1, 0.0)])
@pytest.mark.parametrize("shape", [(10,), (64, 3, 3, 3), (5, 7)])
def test_sgd(backend, ref_lib, nesterov, damp, shape):
    rng = np.random.default_rng(8)
    g, a, m = srnd(rng, *shape), srnd(rng, *shape), srnd(rng, *shape)
    cmd = nnc.CMD_SGD_FORWARD(nesterov, 0.001, 1.0 / 32, 0.0005, 0.9, damp)
    got, want = exec_pair(backend, ref_lib, cmd, nnc.NO_HINT, 0, [g, a, m], [np.zeros(shape, F), np.zeros(shape, F)])
    for x, y in zip(got, want):
        assert tensor_eq(x, y)  # fma contraction may differ by an ulp between compilers


def test_data_transfer_roundtrip(backend):
    rng = np.random.default_rng(9)
    a = srnd(rng, 3, 4, 5)
    L = backend
    src = L.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, 3, 4, 5), a)
    dev = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, 3, 4, 5))
    dev2 = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, 3, 4, 5))
    back = L.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, 3, 4, 5))
    t = nnc.CMD_DATA_TRANSFER_FORWARD()
    assert L.cmd_exec(t, nnc.NO_HINT, 0, [src], [dev]) == 0
    assert L.cmd_exec(t, nnc.NO_HINT, 0, [dev], [dev2]) == 0
    assert L.cmd_exec(t, nnc.NO_HINT, 0, [dev2], [back]) == 0
    assert np.array_equal(back.numpy(), a)


@pytest.mark.parametrize("label_kind", ["f32", "i32", "dense"])
@pytest.mark.parametrize("trim", [(0.0, 1.0), (0.1, 0.9)])
@pytest.mark.parametrize("with_g", [True, False])
def test_softmax_crossentropy(backend, ref_lib, label_kind, trim, with_g):
    rng = np.random.default_rng(10)
    n, c = 6, 37
    a = srnd(rng, n, c, scale=3)
    idx = rng.integers(0, c, n)
    if label_kind == "f32":
        label = idx.astype(F)
    elif label_kind == "i32":
        label = idx.astype(np.int32)
    else:
        label = rnd(rng, n, c)
        label /= label.sum(1, keepdims=True)
    fcmd = nnc.CMD_SOFTMAX_CROSSENTROPY_FORWARD(*trim)
    got, want = exec_pair(backend, ref_lib, fcmd, nnc.NO_HINT, 0, [a, label], [np.zeros(n, F), np.zeros((n, c), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-5, atol=1e-7)
    g = srnd(rng, n) if with_g else None
    bcmd = nnc.CMD_SOFTMAX_CROSSENTROPY_BACKWARD(*trim)
    got, want = exec_pair(backend, ref_lib, bcmd, nnc.NO_HINT, 0, [g, None, None, label, None, want[1]], [np.zeros((n, c), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-7)


def test_softmax_only(backend, ref_lib):
    rng = np.random.default_rng(11)
    a = srnd(rng, 5, 1000, scale=4)
    label = rng.integers(0, 1000, 5).astype(F)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SOFTMAX_CROSSENTROPY_FORWARD(), nnc.NO_HINT, 0, [a, label], [None, np.zeros((5, 1000), F)])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("case", [WINO_CASES[2], WINO_CASES[3]])
def test_conv_winograd_image_slices(backend, ref_lib, case):
    """WINO_SLICE_KB: the Winograd stages run per slice of images (scratch of one slice sized to stay in the Infinity Cache).
    Forced down to one image per slice (and an uneven last slice) here; forward, data and filter gradient (which then
    accumulates its transformed-domain sums across slices) and the fused bias gradient must not change."""
    n, h, w, c, k, border = case
    a, wt, b, hint, oh, ow = _wino_inputs(case)
    g = srnd(np.random.default_rng(5), n, oh, ow, k)
    per_image_kb = 36 * 4 * ((oh + 3) // 4) * ((ow + 3) // 4) * (c + k) / 1024.0
    try:
        for images in (1, 2):
            backend.tune_set("WINO_SLICE_KB", int(per_image_kb * images) + 1)
            cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
            got, want = _wino_pair(backend, ref_lib, cmd, hint, [a, wt, b], [np.zeros((n, oh, ow, k), F)])
            np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
            cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
            got, want = _wino_pair(backend, ref_lib, cmd, hint, [g, a, wt], [np.full_like(a, 3), np.zeros_like(wt), np.zeros(k, F)])
            for i in range(3):
                np.testing.assert_allclose(got[i], want[i], rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(want[i]).max())))
    finally:
        backend.tune_set("WINO_SLICE_KB", 0)


FUSED_CASES = [
    # n, h, w, c, k, border: fused Winograd (algorithm 2): reduction channels % 8 == 0
    (1, 16, 16, 16, 32, (1, 1)),    # exactly one 4x4 tile group, two chunks, one k block
    (2, 13, 13, 32, 32, (1, 1)),    # ragged tiles in a 4x4 group (13 -> 4 tiles, 3 clipped rows / columns), several chunks
    (1, 27, 30, 16, 48, (1, 1)),    # 7 x 8 tiles: the 2x8 group shape, ragged K (48 -> two k blocks, the second half empty)
    (3, 9, 14, 24, 20, (0, 0)),     # no padding, K < 32
    (2, 5, 7, 16, 8, (2, 2)),       # full padding
    (1, 33, 9, 16, 40, (1, 0)),     # tall image: the 8x2 group shape; asymmetric padding
    (5, 20, 20, 16, 64, (1, 1)),    # more groups than one workgroup's four waves, two k blocks
    (9, 30, 30, 24, 96, (1, 1)),    # many work items (9 x 4 groups / 4 x 3 k blocks = 27): every persistent workgroup of the emulator's device walks several
    (2, 12, 17, 64, 64, (1, 1)),    # four pairs of chunks per item (the paired schedule's steady state), two k blocks
]


@pytest.mark.parametrize("grid", [0, 2, 5])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_conv_winograd_fused(backend, ref_lib, case, grid):
    """cmd.algorithm = 2: the fused Winograd kernel (wino_fused.h; LDS-DMA patches, in-register input and output transforms,
    persistent workgroups streaming over work items) for forward and the data gradient (reduction channels >= 16), against the
    reference's direct convolution at 1e-4."""
    n, h, w, c, k, border = case
    a, wt, b, hint, oh, ow = _wino_inputs(case)
    backend.tune_set("WINO_FUSED_GRID", grid)  # 2 / 5 workgroups: each walks several work items (the stream across items, uneven ranges)
    try:
        _fused_check(backend, ref_lib, case, a, wt, b, hint, oh, ow)
    finally:
        backend.tune_set("WINO_FUSED_GRID", 0)


def _fused_check(backend, ref_lib, case, a, wt, b, hint, oh, ow):
    n, h, w, c, k, border = case
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, oh, ow, k), F)], backend=nnc.BACKEND_CPU_REF)
    cmd.algorithm = 2
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.full((n, oh, ow, k), 7, F)])
    assert r1 == 0 and r2 == 0
    assert backend.dll.nnc_mi355x_last_kernel_name().decode() == "conv_fwd_wino_fused"
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
    g = srnd(np.random.default_rng(5), n, oh, ow, k)
    bw = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, bw, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    bw.algorithm = 2
    r1, got = exec_on(backend, nnc.GPU_MEMORY, bw, hint, 0, [g, a, wt], [np.full_like(a, 3), np.zeros_like(wt), np.zeros(k, F)])
    assert r1 == 0 and r2 == 0
    for i in range(3):
        np.testing.assert_allclose(got[i], want[i], rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(want[i]).max())))


WGRAD_FUSED_CASES = [
    # n, h, w, c, k, border: the fused Winograd filter gradient (wino_wgrad_fused.h): C % 64 == 0, K % 32 == 0
    (2, 12, 17, 64, 64, (1, 1)),    # ragged 2 x 4 tile groups (3 x 5 tiles), two k blocks
    (1, 9, 9, 64, 32, (1, 1)),      # fewer tile groups than slices: most slices are empty and write zeros
    (3, 23, 30, 128, 96, (0, 0)),   # no padding; 2 c blocks x 3 k blocks
    (2, 8, 16, 64, 64, (2, 2)),     # full padding: the gradient is larger than the input
    (5, 30, 30, 64, 64, (1, 1)),    # 5 x 4 x 2 = 40 tile groups over 8+ slices: several trips per workgroup, double-buffered stages
    (1, 33, 7, 64, 32, (1, 0)),     # tall, narrow, asymmetric padding
]


@pytest.mark.parametrize("flags", [0, nnc.ACCUMULATE_OUTPUT], ids=["store", "accumulate"])
@pytest.mark.parametrize("case", WGRAD_FUSED_CASES)
def test_conv_wgrad_winograd_fused(backend, ref_lib, case, flags):
    """cmd.algorithm = 2 on the backward row with C % 64 == 0 and K % 32 == 0: the filter gradient with BOTH Winograd transforms in registers (neither
    B^T a B nor G' g G'^T in HBM), tiles as the MFMA reduction dimension, split over tile ranges, slices folded in a fixed order; bias gradient from
    the same pass.  Against the reference's direct loops; under ACCUMULATE the CPU oracle overwrites dbias (conv_cpu_ref.c:262-263) -- see DESIGN.md."""
    n, h, w, c, k, border = case
    a, wt, b, hint, oh, ow = _wino_inputs(case)
    g = srnd(np.random.default_rng(6), n, oh, ow, k)
    dw0, db0 = srnd(np.random.default_rng(7), k, 3, 3, c, scale=0.1), srnd(np.random.default_rng(8), k, scale=0.1)
    bw = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, bw, hint, flags, [g, a, wt], [np.zeros_like(a), dw0.copy(), db0.copy()], backend=nnc.BACKEND_CPU_REF)
    bw.algorithm = 2
    r1, got = exec_on(backend, nnc.GPU_MEMORY, bw, hint, flags, [g, a, wt], [None, dw0.copy(), db0.copy()])
    assert r1 == 0 and r2 == 0
    assert backend.dll.nnc_mi355x_last_kernel_name().decode() == "conv_wgrad_wino_fused"
    scale = max(1.0, float(np.abs(want[1]).max()))
    np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5 * scale)
    want_db = g.sum(axis=(0, 1, 2), dtype=np.float64) + (db0 if flags else 0)
    np.testing.assert_allclose(got[2], want_db, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want_db).max())))


C3_CASES = [
    # n, h, w, k, border[, stride]: 3x3 on THREE input channels -> conv_c3.h (K = 16 / 32 / 64), under the backend's own choice
    (2, 9, 21, 64, (0, 0)),     # VGG-D conv1_1 class: no padding, ragged 16-pixel groups (19 wide)
    (1, 16, 16, 64, (1, 1)),    # BASELINE config 1 class: padding 1
    (3, 7, 40, 32, (1, 1)),
    (2, 5, 6, 16, (2, 2)),      # full padding, rows shorter than one group
    (1, 18, 33, 64, (1, 0)),    # asymmetric padding
    (2, 20, 38, 32, (1, 1), (2, 2)),   # ResNet-50 v1d's stem: stride 2, 32 filters
    (1, 17, 33, 64, (1, 1), (2, 2)),   # odd sizes under stride 2
    (2, 9, 23, 16, (0, 0), (2, 3)),    # different strides per axis, no padding
]


@pytest.mark.parametrize("case", C3_CASES)
def test_conv_first_layer_direct(backend, ref_lib, case):
    n, h, w, k, border = case[:5]
    stride = case[5] if len(case) > 5 else (1, 1)
    rng = np.random.default_rng(9)
    a, wt, b = srnd(rng, n, h, w, 3), srnd(rng, k, 3, 3, 3, scale=1.0 / 27), srnd(rng, k)
    hint = nnc.HINT(stride, border)
    oh, ow = out_hw(h, w, 3, 3, hint)
    fwd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, 3)
    got, want = exec_pair(backend, ref_lib, fwd, hint, 0, [a, wt, b], [np.full((n, oh, ow, k), 7, F)])
    assert backend.dll.nnc_mi355x_last_kernel_name().decode() == "conv_fwd_c3"
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
    g = srnd(rng, n, oh, ow, k)
    bwd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, 3)
    dw0, db0 = srnd(rng, *wt.shape), srnd(rng, k)
    for flags in (0, nnc.ACCUMULATE_OUTPUT):
        got, want = exec_pair(backend, ref_lib, bwd, hint, flags, [g, a, wt], [np.zeros_like(a), dw0.copy(), db0.copy()])
        # (the data gradient runs last; under a stride it is the parity-class form, whose launches are forward contractions)
        assert backend.dll.nnc_mi355x_last_kernel_name().decode() in ("conv_wgrad_c3", "conv_dgrad", "conv_dgrad_wino", "conv_dgrad_wino_fused") + (("conv_fwd",) if stride != (1, 1) else ())
        np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5)
        # the CPU oracle overwrites dbias under ACCUMULATE (see test_conv_backward): the GPU backend being replaced accumulates
        np.testing.assert_allclose(got[2], (db0 + want[2]) if flags else want[2], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("fmt,shape", [("NCHW", (3, 6, 5, 7)), ("NCHW", (2, 8, 8, 8)), ("NHWC", (3, 5, 7, 6)), ("NCHW", (2, 3, 40, 44)),  # (2, 3, 40, 44): planes of 1760 floats, several trips per lane
                                       ("NCHW", (5, 6, 7, 7)), ("NCHW", (3, 5, 14, 14))])  # ResNet's small planes: four planes per wave, 30 / 15 planes (ragged last waves)
def test_batch_norm_with_one_dimensional_statistics(backend, ref_lib, fmt, shape):
    """What ccv_cnnp_batch_norm issues (lib/nnc/ccv_cnnp_model_addons.c:951-986): scale / bias / mean / var / saved tensors of ONE
    dimension (C) against an image tensor; the channel axis is the one the tensor FORMAT names.  The reference's CPU backend
    right-aligns statistics against the data (batch_norm_cpu_ref.c:28-33), so the oracle is run with the same numbers shaped
    (1, C, 1, 1) / (1, 1, 1, C).  (2, 8, 8, 8) NCHW is the trap: W == C, where a right-aligned reading normalises over W.)"""
    rng = np.random.default_rng(11)
    caxis = 1 if fmt == "NCHW" else 3
    C = shape[caxis]
    s4 = tuple(C if k == caxis else 1 for k in range(4))
    axes = tuple(k for k in range(4) if k != caxis)
    x = srnd(rng, *shape, scale=2.0)
    scale, bias = srnd(rng, C) + F(1.5), srnd(rng, C)
    mean, var = srnd(rng, C), rng.random(C, dtype=F) + F(0.5)
    cmd = nnc.CMD_BATCH_NORM_FORWARD(1e-4, 0, 0.9, *axes)

    def run(lib, mem, sshape, backend_id=None):
        r = lambda a: a.reshape(sshape).copy()
        tx, = make_tensors(lib, mem, [x], fmt)
        ts = make_tensors(lib, mem, [r(scale), r(bias), r(mean), r(var)], fmt)
        ty, tsm, tsi = make_tensors(lib, mem, [np.zeros_like(x), np.zeros(sshape, F), np.zeros(sshape, F)], fmt)
        c = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        if backend_id is not None:
            c.backend = backend_id
        assert lib.cmd_exec(c, nnc.NO_HINT, 0, [tx] + ts, [ty, ts[2], ts[3], tsm, tsi]) == 0
        g = srnd(np.random.default_rng(12), *shape)
        tg, = make_tensors(lib, mem, [g], fmt)
        th, tds, tdb = make_tensors(lib, mem, [np.zeros_like(x), np.zeros(sshape, F), np.zeros(sshape, F)], fmt)
        cb = nnc.CMD_BATCH_NORM_BACKWARD(1e-4, 0, 0.9, *axes)
        if backend_id is not None:
            cb.backend = backend_id
        assert lib.cmd_exec(cb, nnc.NO_HINT, 0, [tg] + [None] * 4 + [tx, ts[0]] + [None] * 6 + [tsm, tsi], [th, tds, tdb]) == 0
        return [t.numpy().reshape(-1) if t.numpy().ndim != 4 or t.numpy().shape != x.shape else t.numpy() for t in (ty, ts[2], ts[3], tsm, tsi, th, tds, tdb)]

    got = run(backend, nnc.GPU_MEMORY, (C,))
    want = run(ref_lib, nnc.CPU_MEMORY, s4, nnc.BACKEND_CPU_REF)
    for a, b, what in zip(got, want, ("y", "mean", "var", "saved_mean", "saved_inv_std", "h", "dscale", "dbias")):
        np.testing.assert_allclose(a.reshape(-1), b.reshape(-1), rtol=2e-4, atol=2e-5, err_msg=what)


@pytest.mark.parametrize("case", [(3, 8, 6, 6, 12, True), (2, 16, 4, 8, 8, False), (2, 64, 14, 14, 128, True), (16, 64, 14, 14, 160, True), (68, 8, 6, 6, 12, False), (8, 256, 14, 14, 128, True)],
                         ids=["8to12", "16to8", "64to128", "64to160-batch16-one-image-per-xcd", "8to12-batch68-ragged-last-round", "256to128-batch8-fp32-on-the-bf16-pipe"])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_conv_1x1_on_nchw_tensors_without_layout_passes(backend, ref_lib, case, dtype):
    """ResNet's bottleneck convolutions as the reference's trainer issues them (NCHW, 1x1, stride 1: bin/nnc/imagenet.c): forward,
    data gradient, filter gradient and bias gradient run as GEMMs over the NCHW tensors where they lie (conv1x1_nchw_*), no
    transposes -- checked against the oracle (CPU_REF convolution forward on NCHW; its backward is NHWC-only, so the oracle's
    backward runs on transposed copies) and by kernel name."""
    n, c, h, w_, k, with_bias = case
    rng = np.random.default_rng(21)
    half = dtype == "f16"
    T = np.float16 if half else F
    a = srnd(rng, n, c, h, w_).astype(T)
    wt = srnd(rng, k, c, 1, 1, scale=1.0 / np.sqrt(c)).astype(T)
    bias = srnd(rng, k).astype(T)
    g = srnd(rng, n, k, h, w_, scale=0.5).astype(T)
    tol = dict(rtol=5e-3, atol=5e-3) if half else dict(rtol=1e-4, atol=1e-5)
    names = []

    def gpu(cmd, ins, outs):
        backend.profile_enable(1)
        r, res = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.HINT((1, 1), (0, 0)), 0, ins, outs, fmt="NCHW")
        backend.stream_wait(None)
        names.extend(x[0] for x in backend.profile_records())
        backend.profile_enable(0)
        assert r == 0
        return res
    b, = gpu(nnc.CMD_CONVOLUTION_FORWARD(1, k, 1, 1, c), [a, wt] + ([bias] if with_bias else []), [np.zeros((n, k, h, w_), T)])
    hh, dw, db = gpu(nnc.CMD_CONVOLUTION_BACKWARD(1, k, 1, 1, c), [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, T)])
    a32, w32, g32 = a.astype(F), wt.astype(F).reshape(k, c), g.astype(F)
    want_b = np.einsum("kc,nchw->nkhw", w32, a32) + (bias.astype(F)[None, :, None, None] if with_bias else 0)
    want_h = np.einsum("kc,nkhw->nchw", w32, g32)
    want_dw = np.einsum("nkhw,nchw->kc", g32, a32).reshape(wt.shape)
    want_db = g32.sum(axis=(0, 2, 3))
    # the einsum is pinned against the reference's own forward on the same NCHW tensors
    r, (ref_b,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_CONVOLUTION_FORWARD(1, k, 1, 1, c), nnc.HINT((1, 1), (0, 0)), 0, [a32, wt.astype(F)] + ([bias.astype(F)] if with_bias else []), [np.zeros((n, k, h, w_), F)], fmt="NCHW", backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    np.testing.assert_allclose(want_b, ref_b, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b.astype(F), want_b, **tol)
    np.testing.assert_allclose(hh.astype(F), want_h, **tol)
    np.testing.assert_allclose(dw.astype(F), want_dw, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(np.abs(want_dw).max())))
    np.testing.assert_allclose(db.astype(F), want_db, rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(np.abs(want_db).max())))
    if n >= 16:  # round 6: the batch entries on ONE grid dimension, an image per XCD (gemm_batch_xcd_map) -- the same arithmetic as with the images on grid z: EQUAL
        backend.tune_set("GEMM_BATCH_XCD", 0)
        try:
            b0, = gpu(nnc.CMD_CONVOLUTION_FORWARD(1, k, 1, 1, c), [a, wt] + ([bias] if with_bias else []), [np.zeros((n, k, h, w_), T)])
            h0, dw0, db0 = gpu(nnc.CMD_CONVOLUTION_BACKWARD(1, k, 1, 1, c), [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, T)])
        finally:
            backend.tune_set("GEMM_BATCH_XCD", 1)
        assert np.array_equal(b0, b) and np.array_equal(h0, hh) and np.array_equal(dw0, dw) and np.array_equal(db0, db)
    native = (h * w_) % 4 == 0 and c % 4 == 0 and k % 4 == 0
    if native:
        assert any(x.startswith("conv1x1_nchw_fwd") for x in names) and any(x.startswith("conv1x1_nchw_dgrad") for x in names) and any(x.startswith("conv1x1_nchw_wgrad") for x in names), names


STRIDE2_CASES = [
    # n, h, w, c, k, kh, kw, border
    (2, 12, 13, 8, 12, 3, 3, (1, 1)),    # ResNet stage transition shape class: odd and even extents
    (2, 11, 10, 8, 8, 3, 3, (0, 0)),     # no border: the even positions get two taps per axis
    (1, 14, 15, 4, 8, 7, 7, (3, 3)),     # 7 x 7 (stem): 3 + 4 taps per axis
    (2, 9, 8, 5, 6, 2, 2, (0, 0)),       # 2 x 2: exactly one tap per class; scalar scatter (C % 4 != 0)
    (1, 10, 10, 4, 4, 4, 4, (1, 1)),     # even filter
    (1, 9, 9, 4, 4, 3, 5, (1, 2)),       # different extents per axis
]


@pytest.mark.parametrize("case", STRIDE2_CASES, ids=[str(c) for c in STRIDE2_CASES])
def test_conv_stride2_data_gradient_by_parity_classes(backend, ref_lib, case):
    """The data gradient of a stride-2 convolution as four dense stride-1 correlations (one per parity of the input position) +
    an interleave (cmd_conv.cpp: conv_dgrad_parity) against the oracle; the launch records show forward-GEMM launches inside the
    backward command, i.e. the parity path and not the strided im2col walk ran."""
    n, h, w, c, k, kh, kw, border = case
    rng = np.random.default_rng(15)
    hint = nnc.HINT((2, 2), border)
    oh, ow = out_hw(h, w, kh, kw, hint)
    a, wt, g = srnd(rng, n, h, w, c), srnd(rng, k, kh, kw, c, scale=1.0 / (kh * kw * c)), srnd(rng, n, oh, ow, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, kh, kw, c)
    backend.profile_enable(1)
    try:
        got, want = exec_pair(backend, ref_lib, cmd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)])
        backend.stream_wait(None)
        names = [r[0] for r in backend.profile_records()]
    finally:
        backend.profile_enable(0)
    assert any(x.startswith("conv_fwd") for x in names) and not any(x.startswith("conv_dgrad") for x in names), names
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=2e-5)


FUSE_RELU_CASES = [
    # n, h, w, c, k, border, algorithm, fmt
    (2, 13, 14, 16, 24, 1, 0xff, "NHWC"),   # the backend's choice
    (2, 13, 14, 16, 24, 1, 0, "NHWC"),      # implicit GEMM: no fused epilogue, one more pass
    (2, 13, 14, 16, 24, 1, 1, "NHWC"),      # Winograd via HBM: the output transform rectifies
    (2, 13, 14, 16, 32, 1, 2, "NHWC"),      # fused Winograd: its epilogue does
    (2, 12, 11, 3, 64, 0, 0xff, "NHWC"),    # 3-channel kernel
    (2, 9, 9, 8, 8, 1, 0xff, "NCHW"),       # staged layouts
]


@pytest.mark.parametrize("case", FUSE_RELU_CASES, ids=[str(c) for c in FUSE_RELU_CASES])
def test_conv_forward_with_fused_relu(backend, ref_lib, case):
    """cmd.algorithm = NNC_MI355X_CONV_ALGO_FUSE_RELU | a: the command writes max(0, conv + bias) -- the oracle's convolution followed
    by the oracle's ReLU, whichever kernel ran."""
    n, h, w, c, k, border, algo, fmt = case
    rng = np.random.default_rng(21)
    a, wt, b = srnd(rng, n, h, w, c), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), srnd(rng, k, scale=0.05)
    hint = nnc.HINT((1, 1), (border, border))
    oh, ow = out_hw(h, w, 3, 3, hint)
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    _, (conv,) = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, oh, ow, k), F)], backend=nnc.BACKEND_CPU_REF)
    want = np.maximum(conv, 0)
    fused = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    fused.algorithm = nnc.CONV_ALGO_FUSE_RELU | algo
    t = (lambda x: np.ascontiguousarray(x.transpose(0, 3, 1, 2))) if fmt == "NCHW" else (lambda x: x)
    wt_l = np.ascontiguousarray(wt.transpose(0, 3, 1, 2)) if fmt == "NCHW" else wt
    r, (got,) = exec_on(backend, nnc.GPU_MEMORY, fused, hint, 0, [t(a), wt_l, b], [t(np.full((n, oh, ow, k), -7, F))], fmt)
    assert r == 0
    assert (got >= 0).all() and (got == 0).any()
    np.testing.assert_allclose(got, t(want), rtol=1e-4, atol=1e-5)


POOL_RELU_BACK_CASES = [
    # h, w, c, window, stride, fmt
    (13, 13, 8, 3, 2, "NHWC"),   # VGG-D's overlapping windows, 16-byte lanes
    (13, 12, 5, 3, 2, "NHWC"),   # scalar lanes
    (12, 12, 8, 2, 2, "NHWC"),   # windows that tile the map
    (9, 9, 4, 2, 2, "NCHW"),     # tiling, one row / column left over
    (11, 10, 3, 3, 2, "NCHW"),
]


@pytest.mark.parametrize("case", POOL_RELU_BACK_CASES, ids=[str(c) for c in POOL_RELU_BACK_CASES])
def test_max_pool_backward_with_relu_backward_folded_in(backend, ref_lib, case):
    """cmd.algorithm = NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD: the command's result is the oracle's MAX_POOL_BACKWARD followed by the
    oracle's RELU_BACKWARD on the pooled map (a rectified map: about half of it zeros, ties included)."""
    h, w, c, k, s, fmt = case
    rng = np.random.default_rng(33)
    a = np.maximum(srnd(rng, 1, h, w, c), 0)
    a[rng.random(a.shape) < 0.6] = 0  # sparse enough for whole windows of zeros
    hint = nnc.HINT((s, s), (0, 0))
    oh, ow = out_hw(h, w, k, k, hint)
    g = srnd(rng, 1, oh, ow, c)
    _, (b,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_MAX_POOL_FORWARD(k, k), hint, 0, [a], [np.zeros((1, oh, ow, c), F)], backend=nnc.BACKEND_CPU_REF)
    _, (hp,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_MAX_POOL_BACKWARD(k, k), hint, 0, [g, a, b], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    _, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_RELU_BACKWARD(), nnc.NO_HINT, 0, [hp, None, a], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    assert (want != hp).any()  # the mask matters: windows whose maximum is 0 route gradient to zeros
    cmd = nnc.CMD_MAX_POOL_BACKWARD(k, k)
    cmd.algorithm = nnc.POOL_ALGO_FUSE_RELU_BACKWARD
    t = (lambda x: np.ascontiguousarray(x.transpose(0, 3, 1, 2))) if fmt == "NCHW" else (lambda x: x)
    r, (got,) = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, [t(g), t(a), t(b)], [t(np.full_like(a, 5))], fmt)
    assert r == 0
    assert np.array_equal(got, t(want))


CONV_RELU_BACK_CASES = [
    # n, h, w, c, k, border, algorithm, fmt
    (2, 13, 14, 32, 16, 1, 2, "NHWC"),     # fused Winograd data gradient: mask bits packed in the epilogue's order, 2 k-blocks... of c
    (3, 18, 21, 64, 16, 1, 2, "NHWC"),     # several tile groups per image, ragged edges, 2 channel blocks of the gradient
    (2, 13, 14, 16, 24, 1, 1, "NHWC"),     # Winograd via HBM: the output transform reads the map
    (2, 13, 14, 16, 24, 1, 0, "NHWC"),     # implicit GEMM: one more pass
    (2, 13, 14, 16, 24, 1, 0xff, "NHWC"),
    (2, 12, 12, 8, 8, 0, 0xff, "NHWC"),
    (2, 9, 9, 8, 8, 1, 0xff, "NCHW"),
]


@pytest.mark.parametrize("case", CONV_RELU_BACK_CASES, ids=[str(c) for c in CONV_RELU_BACK_CASES])
def test_conv_backward_with_relu_backward_folded_in(backend, ref_lib, case):
    """NNC_MI355X_CONV_ALGO_FUSE_RELU on CONVOLUTION_BACKWARD: h is the oracle's data gradient followed by the oracle's RELU_BACKWARD
    on the forward input a (a rectified map); dw and dbias are the plain ones."""
    n, h, w, c, k, border, algo, fmt = case
    rng = np.random.default_rng(41)
    a = np.maximum(srnd(rng, n, h, w, c), 0)
    wt, hint = srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), nnc.HINT((1, 1), (border, border))
    oh, ow = out_hw(h, w, 3, 3, hint)
    g = srnd(rng, n, oh, ow, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    _, plain = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    _, (want_h,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_RELU_BACKWARD(), nnc.NO_HINT, 0, [plain[0], None, a], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    assert (want_h != plain[0]).any()
    fused = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    fused.algorithm = nnc.CONV_ALGO_FUSE_RELU | algo
    t = (lambda x: np.ascontiguousarray(x.transpose(0, 3, 1, 2))) if fmt == "NCHW" else (lambda x: x)
    r, got = exec_on(backend, nnc.GPU_MEMORY, fused, hint, 0, [t(g), t(a), t(wt)], [t(np.full_like(a, 3)), t(np.zeros_like(wt)), np.zeros(k, F)], fmt)
    assert r == 0
    if algo in (1, 2) and hasattr(backend.dll, "nnc_mi355x_last_kernel_name"):
        assert backend.dll.nnc_mi355x_last_kernel_name().decode() == ("conv_dgrad_wino_fused" if algo == 2 else "conv_dgrad_wino")
    tol = lambda ref: dict(rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    np.testing.assert_allclose(got[0], t(want_h), **tol(want_h))
    assert np.array_equal(got[0] == 0, t(want_h) == 0) or np.abs(got[0][(got[0] == 0) != (t(want_h) == 0)]).max() < 1e-6
    np.testing.assert_allclose(got[1], t(plain[1]), **tol(plain[1]))
    np.testing.assert_allclose(got[2], plain[2], **tol(plain[2]))
    # without a data gradient asked for the bit means nothing
    r, got = exec_on(backend, nnc.GPU_MEMORY, fused, hint, 0, [t(g), t(a), t(wt)], [None, t(np.zeros_like(wt)), np.zeros(k, F)], fmt)
    assert r == 0
    np.testing.assert_allclose(got[1], t(plain[1]), **tol(plain[1]))


# ---- the contraction epilogue's two ways out (mfma_gemm.h "epilogues"): one element per lane, or the block tile staged through LDS and written as 16-byte
# (8-byte for halves) row segments.  Same arithmetic in the same order: bit-identical, with bias / alpha / accumulate, ragged M, batches, split-K slabs.
VEC_EPI_CASES = [
    # (name, command builder, hint, flags, inputs, outputs, fmt)
    ("gemm 130x100 . 260x100^T + bias", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), None, 0, [(130, 100), (260, 100), (260,)], [(130, 260)], "NHWC"),
    ("gemm accumulate", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE), None, nnc.ACCUMULATE_OUTPUT, [(70, 64), (64, 72)], [(70, 72)], "NHWC"),
    ("gemm split-K", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), None, 0, [(8, 4096), (16, 4096), (16,)], [(8, 16)], "NHWC"),
    ("gemm K 4096 (the half buffer-load kernel)", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), None, 0, [(130, 4096), (136, 4096), (136,)], [(130, 136)], "NHWC"),
    ("gemm K 4096 n-contiguous B", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE), None, nnc.ACCUMULATE_OUTPUT, [(72, 4096), (4096, 136)], [(72, 136)], "NHWC"),
    ("gemm batched", lambda: nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE), None, 0, [(3, 36, 20), (3, 20, 8)], [(3, 36, 8)], "NHWC"),
    ("conv 3x3 nhwc", lambda: nnc.CMD_CONVOLUTION_FORWARD(1, 24, 3, 3, 8), ((1, 1), (1, 1)), 0, [(2, 9, 9, 8), (24, 3, 3, 8), (24,)], [(2, 9, 9, 24)], "NHWC"),
    ("conv 1x1 nchw", lambda: nnc.CMD_CONVOLUTION_FORWARD(1, 72, 1, 1, 32), ((1, 1), (0, 0)), 0, [(3, 32, 6, 6), (72, 32, 1, 1), (72,)], [(3, 72, 6, 6)], "NCHW"),
    ("conv 1x1 nchw backward", lambda: nnc.CMD_CONVOLUTION_BACKWARD(1, 72, 1, 1, 32), ((1, 1), (0, 0)), 0, [(3, 72, 6, 6), (3, 32, 6, 6), (72, 32, 1, 1)], [(3, 32, 6, 6), (72, 32, 1, 1), (72,)], "NCHW"),
    ("conv 3x3 nchw (half: the planar epilogue)", lambda: nnc.CMD_CONVOLUTION_FORWARD(1, 72, 3, 3, 64), ((1, 1), (1, 1)), 0, [(3, 64, 10, 10), (72, 64, 3, 3), (72,)], [(3, 72, 10, 10)], "NCHW"),
    ("conv 3x3 nchw backward (half: the planar epilogue)", lambda: nnc.CMD_CONVOLUTION_BACKWARD(1, 72, 3, 3, 64), ((1, 1), (1, 1)), 0, [(3, 72, 10, 10), (3, 64, 10, 10), (72, 64, 3, 3)], [(3, 64, 10, 10), (72, 64, 3, 3), (72,)], "NCHW"),
    ("conv 1x1 nchw backward accumulate", lambda: nnc.CMD_CONVOLUTION_BACKWARD(1, 72, 1, 1, 32), ((1, 1), (0, 0)), nnc.ACCUMULATE_OUTPUT, [(3, 72, 6, 6), (3, 32, 6, 6), (72, 32, 1, 1)], [(3, 32, 6, 6), (72, 32, 1, 1), (72,)], "NCHW"),
]


@pytest.mark.parametrize("case", VEC_EPI_CASES, ids=[c[0] for c in VEC_EPI_CASES])
@pytest.mark.parametrize("dtype", [np.float32, np.float16], ids=["f32", "f16"])
def test_vector_epilogue_is_bit_identical_to_the_scalar_one(backend, ref_lib, case, dtype):
    name, mk, hint, flags, in_shapes, out_shapes, fmt = case
    rng = np.random.default_rng(11)
    ins = [srnd(rng, *s, scale=0.05 if ("4096" in name or "planar" in name) else 1.0).astype(dtype) for s in in_shapes]
    outs = [srnd(rng, *s).astype(dtype) for s in out_shapes]
    h = nnc.HINT(*hint) if hint else nnc.NO_HINT
    res = {}
    try:
        for vec in (1, 0):
            backend.tune_set("GEMM_VEC_EPILOGUE", vec)
            r, got = exec_on(backend, nnc.GPU_MEMORY, mk(), h, flags, ins, outs, fmt)
            assert r == 0
            res[vec] = got
    finally:
        backend.tune_set("GEMM_VEC_EPILOGUE", 1)
    for a, b in zip(res[1], res[0]):
        assert np.array_equal(a.view(np.uint32 if dtype == np.float32 else np.uint16), b.view(np.uint32 if dtype == np.float32 else np.uint16)), name
    # and both are right: the oracle on the same values in fp32 (the reference's CPU backward pass takes NHWC-format filters only: the NCHW backward cases are
    # held to the oracle by test_conv1x1_nchw_* / test_resnet_block.py)
    if "nchw backward" in name or "planar" in name:  # (... nor NCHW-format 3 x 3 filters: tests/test_half.py holds those to the oracle)
        return
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, mk(), h, flags, [x.astype(F) for x in ins], [x.astype(F) for x in outs], fmt, backend=nnc.BACKEND_CPU_REF)
    assert r2 == 0
    tol = 1e-4 if dtype == np.float32 else 2e-2
    for a, b in zip(res[1], want):
        np.testing.assert_allclose(a.astype(F), b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max())))
