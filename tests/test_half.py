"""CCV_16F datapath (BASELINE config 5): half-precision tensors through the command interface.

  * contraction commands (GEMM forward / backward, convolution forward / backward) run on the half-precision MFMA core
    (mfma_gemm_f16.h, v_mfma_f32_32x32x16_f16, fp32 accumulation) when every tensor is CCV_16F and readable in 4-element chunks;
  * every other row, and the contraction rows for mixed types / odd strides, runs its fp32 kernels on fp32 images of the half
    tensors (half_stage.cpp).
Oracle: the reference's CPU backend in fp32 on the SAME half-rounded inputs; the result must agree with the oracle's, rounded
to half, within the tolerance the reference's own half-precision GPU tests use (test/int/nnc/cudnn.tests.c:196 5e-3 on values of
order 1, cublas.tests.c half cases 1e-3 .. 5e-3): |got - want| <= 5e-3 * max(1, |want|max) elementwise.
"""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_on

F, H = np.float32, np.float16


def hrnd(rng, *shape, scale=1.0):
    return ((rng.random(shape, dtype=F) - 0.5) * 2 * scale).astype(H)


def _pair(L, ref, cmd, hint, flags, inputs, outputs):
    """inputs / outputs in half (numpy float16) -> (backend result in half, oracle result in fp32 from the same values)"""
    r1, got = exec_on(L, nnc.GPU_MEMORY, cmd, hint, flags, inputs, outputs)
    up = lambda xs: [None if x is None else (x.astype(F) if x.dtype == H else x) for x in xs]
    r2, want = exec_on(ref, nnc.CPU_MEMORY, cmd, hint, flags, up(inputs), up(outputs), backend=nnc.BACKEND_CPU_REF)
    assert r2 == 0 and r1 == 0, (r1, r2)
    return got, want


def _close(got, want, tol=5e-3):
    assert got.dtype == H
    g, w = got.astype(np.float64), want.astype(np.float64)
    bound = tol * max(1.0, float(np.abs(w).max()))
    err = float(np.abs(g - w).max())
    assert err <= bound, "max |diff| %.4g > %.4g at %s" % (err, bound, np.unravel_index(int(np.abs(g - w).argmax()), g.shape))


GEMM_H = [
    # (a shape, w shape, transpose_a, transpose_b, bias, native)   native: every operand readable in 4-element chunks
    ((8, 32), (12, 32), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True, True),        # cnnp dense layer, k-contiguous x k-contiguous
    ((64, 128), (128, 96), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, True, True),        # k-contiguous x row-contiguous (transposed into LDS)
    ((40, 36), (40, 52), nnc.TRANSPOSE(0, 1), nnc.NO_TRANSPOSE, False, True),      # row-contiguous x row-contiguous
    ((132, 200), (260, 200), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True, True),   # several 128 x 128 tiles, ragged
    ((8, 4096), (16, 4096), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True, True),    # split-K
    ((3, 8, 12), (3, 12, 16), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, False, True),    # batched
    # whole K-steps, 128 x 128 tiles, K >= 4096: the buffer-load kernel (mfma_gemm_f16_buf.h); a row-contiguous operand goes through the LDS transpose read
    ((136, 4096), (264, 4096), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True, "buf"),   # k-contiguous x k-contiguous, ragged tiles
    ((136, 4096), (4096, 196), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, True, "buf"),      # k-contiguous x row-contiguous; 196 columns: the last 8-row chunk straddles the end of a row
    ((4096, 136), (4096, 200), nnc.TRANSPOSE(0, 1), nnc.NO_TRANSPOSE, False, "buf"),  # row-contiguous x row-contiguous
    ((4096, 264), (136, 4096), nnc.TRANSPOSE(0, 1), nnc.TRANSPOSE(0, 1), False, "buf"),  # row-contiguous x k-contiguous
    ((5, 3), (3, 7), nnc.NO_TRANSPOSE, nnc.NO_TRANSPOSE, True, False),             # odd sizes: fp32 core on fp32 images
    ((6, 18), (9, 18), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1), True, False),
]


def _t(t, nd):
    return t if t[0] == t[1] or nd == 2 else (nd - 2, nd - 1)


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


def _kernel_records(L, fn):
    L.profile_enable(1)
    try:
        fn()
        L.stream_wait(None)
        return [r[0] for r in L.profile_records()]
    finally:
        L.profile_enable(0)


@pytest.mark.parametrize("case", GEMM_H, ids=[str(c[:2]) for c in GEMM_H])
def test_gemm_forward_half(backend, ref_lib, case):
    ashape, wshape, ta, tb, bias, native = case
    rng = np.random.default_rng(0)
    scale = 1.0 / np.sqrt(ashape[-1] if ta[0] == ta[1] else ashape[-2])
    a, w = hrnd(rng, *ashape), hrnd(rng, *wshape, scale=4 * scale)
    ta, tb = _t(ta, len(ashape)), _t(tb, len(wshape))
    bshape = _gemm_shapes(ashape, wshape, ta, tb)
    ins = [a, w] + ([hrnd(rng, bshape[-1])] if bias else [])
    res = {}
    names = _kernel_records(backend, lambda: res.update(r=_pair(backend, ref_lib, nnc.CMD_GEMM_FORWARD(ta, tb), nnc.NO_HINT, 0, ins, [np.zeros(bshape, H)])))
    got, want = res["r"]
    _close(got[0], want[0])
    assert any("mfma_gemm_f16" in n for n in names) == bool(native), names
    if native == "buf":
        assert any("mfma_gemm_f16_buf_kernel" in n for n in names), names


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("layout", ["kc x kc", "nc x nc"])
def test_gemm_half_buffer_kernel_shapes(backend, ref_lib, mode, layout):
    """The other shapes of the buffer-load kernel (mfma_gemm_f16_buf.h), forced through the tuning key: the 256 x 256 x 64 tile (mode 3) and K-steps of 32 (mode 4)."""
    rng = np.random.default_rng(5)
    M, N, K = 264, 324, 4096
    if layout == "kc x kc":
        a, w, ta, tb = hrnd(rng, M, K), hrnd(rng, N, K, scale=4.0 / np.sqrt(K)), nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)
    else:
        a, w, ta, tb = hrnd(rng, K, M), hrnd(rng, K, N, scale=4.0 / np.sqrt(K)), nnc.TRANSPOSE(0, 1), nnc.NO_TRANSPOSE
    backend.tune_set("GEMM_BUFFER_LOADS", mode)
    try:
        res = {}
        names = _kernel_records(backend, lambda: res.update(r=_pair(backend, ref_lib, nnc.CMD_GEMM_FORWARD(ta, tb), nnc.NO_HINT, 0, [a, w, hrnd(rng, N)], [np.zeros((M, N), H)])))
    finally:
        backend.tune_set("GEMM_BUFFER_LOADS", 1)
    got, want = res["r"]
    _close(got[0], want[0])
    assert any((("256 x 256 x 64" if layout == "kc x kc" else "256 x 256 x 32") if mode == 3 else "128 x 128 x 32") in n for n in names), names


GEMM_H_BACK = [c for c in GEMM_H if c[5] != "buf"] + [c for c in GEMM_H if c[5] == "buf"][:2]  # (the reference's K = 4096 products are what takes the time here)


@pytest.mark.parametrize("case", GEMM_H_BACK, ids=[str(c[:2]) for c in GEMM_H_BACK])
@pytest.mark.parametrize("flags", [0, nnc.ACCUMULATE_OUTPUT])
def test_gemm_backward_half(backend, ref_lib, case, flags):
    ashape, wshape, ta, tb, bias, native = case
    rng = np.random.default_rng(1)
    a, w = hrnd(rng, *ashape), hrnd(rng, *wshape)
    ta, tb = _t(ta, len(ashape)), _t(tb, len(wshape))
    bshape = _gemm_shapes(ashape, wshape, ta, tb)
    g = hrnd(rng, *bshape, scale=1.0 / np.sqrt(max(bshape[-1], bshape[-2])))
    outs = [hrnd(rng, *ashape), hrnd(rng, *wshape)] + ([hrnd(rng, bshape[-1])] if bias else [])
    got, want = _pair(backend, ref_lib, nnc.CMD_GEMM_BACKWARD(ta, tb), nnc.NO_HINT, flags, [g, a, w], outs)
    for x, y in zip(got, want):
        _close(x, y)


def test_relu_and_add_half_through_fp32_images(backend, ref_lib):
    rng = np.random.default_rng(2)
    a, b = hrnd(rng, 3, 5, 7, 6), hrnd(rng, 3, 5, 7, 6)
    got, want = _pair(backend, ref_lib, nnc.CMD_RELU_FORWARD(), nnc.NO_HINT, 0, [a], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0].astype(H))  # exact: max(a, 0) of a half is a half
    got, want = _pair(backend, ref_lib, nnc.CMD_ADD_FORWARD(0.5, 0.25), nnc.NO_HINT, 0, [a, b], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0].astype(H))  # one rounding, of the fp32 result


def test_sgd_mixed_precision(backend, ref_lib):
    """fp16 gradient, fp32 parameters and momentum (lib/nnc/cmd/sgd/gpu/ccv_nnc_sgd_gpu_ref.cu:13-100, the mixed variants)."""
    rng = np.random.default_rng(3)
    g = hrnd(rng, 1000)
    a, m = (rng.random(1000, dtype=F) - 0.5).astype(F), (rng.random(1000, dtype=F) - 0.5).astype(F)
    cmd = nnc.CMD_SGD_FORWARD(0, 0.01, 0.5, 0.0005, 0.9, 0.9)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [g, a, m], [np.zeros_like(a), np.zeros_like(m)])
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [g.astype(F), a, m], [np.zeros_like(a), np.zeros_like(m)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    for x, y in zip(got, want):
        assert x.dtype == F
        np.testing.assert_allclose(x, y, rtol=1e-6, atol=1e-7)


def test_half_view_output_keeps_what_it_does_not_write(backend, ref_lib):
    """An output VIEW in half precision: elements outside the view must survive the round trip through the fp32 image."""
    L = backend
    rng = np.random.default_rng(4)
    base = hrnd(rng, 6, 10)
    a = hrnd(rng, 6, 4)
    bt = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, base.shape, 0), base)
    at = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, a.shape, 0), a)
    view = bt.view((6, 4), (10, 1), 3)
    assert L.cmd_exec(nnc.CMD_ADD_FORWARD(1, 1), nnc.NO_HINT, 0, [at, at], [view]) == 0
    out = bt.numpy()
    want = base.copy()
    want[:, 3:7] = (a.astype(F) * 2).astype(H)
    assert np.array_equal(out, want)


def test_half_sibling_view_outputs_do_not_clobber_each_other(backend):
    """two output VIEWS of one half-precision parent (channel-concatenated branches): each is written back element by element
    through its own strides.  Writing a view back as its whole span rewrote the sibling's elements with what had been loaded before the
    command ran -- here the second write-back would have undone the first."""
    L = backend
    rng = np.random.default_rng(14)
    g = hrnd(rng, 5, 6)
    parent = hrnd(rng, 5, 12)
    gt = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, g.shape, 0), g)
    pt = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, parent.shape, 0), parent)
    left, right = pt.view((5, 6), (12, 1), 0), pt.view((5, 6), (12, 1), 6)
    assert L.cmd_exec(nnc.CMD_ADD_BACKWARD(0.5, 2.0), nnc.NO_HINT, 0, [gt], [left, right]) == 0
    out = pt.numpy()
    want = np.concatenate([(g.astype(F) * F(0.5)).astype(H), (g.astype(F) * F(2.0)).astype(H)], axis=1)
    assert np.array_equal(out, want)
    # one view at a time, the other half untouched
    pt2 = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, parent.shape, 0), parent)
    assert L.cmd_exec(nnc.CMD_ADD_FORWARD(1, 2), nnc.NO_HINT, 0, [gt, gt], [pt2.view((5, 6), (12, 1), 6)]) == 0
    out2 = pt2.numpy()
    assert np.array_equal(out2[:, :6], parent[:, :6]) and np.array_equal(out2[:, 6:], (g.astype(F) * F(3.0)).astype(H))


CONV_H = [
    # n, h, w, c, k, kh, kw, stride, border, groups, native
    (2, 9, 10, 8, 16, 3, 3, (1, 1), (1, 1), 1, True),
    (1, 12, 11, 32, 20, 3, 3, (1, 1), (1, 1), 1, True),     # chunk-major K order (C % 32 == 0)
    (2, 11, 9, 12, 8, 5, 5, (2, 2), (2, 2), 1, True),       # stride 2: strided dgrad loader
    (2, 7, 7, 16, 24, 1, 1, (1, 1), (0, 0), 2, True),       # 1x1, two groups
    (8, 7, 7, 96, 160, 1, 1, (1, 1), (0, 0), 1, True),      # 1x1, one group: two plain matrices (conv_pointwise), 392 pixels -- ResNet-50's 7 x 7 maps
    (3, 16, 16, 64, 64, 3, 3, (1, 1), (1, 1), 1, True),     # several K-steps, 64 x 64 tiles
    (2, 9, 9, 3, 8, 3, 3, (1, 1), (1, 1), 1, False),        # 3 input channels: fp32 kernels on fp32 images
]


@pytest.mark.parametrize("case", CONV_H, ids=[str(c[:7]) for c in CONV_H])
def test_conv_forward_backward_half(backend, ref_lib, case):
    n, h, w_, c, k, kh, kw, stride, border, groups, native = case
    rng = np.random.default_rng(7)
    a = hrnd(rng, n, h, w_, c)
    wt = hrnd(rng, k, kh, kw, c // groups, scale=2.0 / np.sqrt(kh * kw * c // groups))
    bias = hrnd(rng, k)
    hint = nnc.HINT(stride, border)
    oh = (h + 2 * border[0] - kh) // stride[0] + 1
    ow = (w_ + 2 * border[1] - kw) // stride[1] + 1
    res = {}
    names = _kernel_records(backend, lambda: res.update(r=_pair(backend, ref_lib, nnc.CMD_CONVOLUTION_FORWARD(groups, k, kh, kw, c // groups), hint, 0, [a, wt, bias], [np.zeros((n, oh, ow, k), H)])))
    got, want = res["r"]
    _close(got[0], want[0])
    assert any("mfma_gemm_f16_kernel" in x for x in names) == native, names
    g = hrnd(rng, n, oh, ow, k, scale=1.0 / np.sqrt(oh * ow))
    for flags in (0, nnc.ACCUMULATE_OUTPUT):
        outs = [hrnd(rng, n, h, w_, c), hrnd(rng, k, kh, kw, c // groups), hrnd(rng, k)]
        got, want = _pair(backend, ref_lib, nnc.CMD_CONVOLUTION_BACKWARD(groups, k, kh, kw, c // groups), hint, flags, [g, a, wt], outs)
        _close(got[0], want[0])
        _close(got[1], want[1])
        if not flags:  # (the CPU oracle overwrites dbias under ACCUMULATE_OUTPUT, conv_cpu_ref.c:262-263; the GPU backend being replaced accumulates)
            _close(got[2], want[2])


@pytest.mark.parametrize("case", [(2, 10, 10, 16, 24, 3, 3, (1, 1), (1, 1)), (3, 8, 8, 3, 8, 3, 3, (1, 1), (1, 1)), (2, 9, 9, 8, 8, 5, 5, (2, 2), (2, 2)),
                                  (2, 10, 10, 64, 72, 3, 3, (1, 1), (1, 1)), (2, 9, 9, 64, 64, 5, 5, (2, 2), (2, 2)),
                                  (3, 8, 16, 64, 72, 3, 3, (1, 1), (1, 1)), (264, 8, 16, 64, 64, 3, 3, (1, 1), (1, 1))],
                         ids=["3x3", "3x3-c3", "5x5-s2", "3x3-c64-f16", "5x5-s2-c64-f16", "3x3-c64-f16-bias-sums-in-the-layout-pass", "3x3-c64-f16-bias-sums-two-level-fold"])
def test_conv_half_nchw_through_converting_transposes(backend, ref_lib, case, request):
    """CCV_16F tensors and filters in NCHW, kernel larger than 1 x 1 -- the CIFAR-10 / ImageNet trainers' fp16 mode.  One
    converting transpose per tensor feeds the fp32 NHWC kernels (no fp32 image of the NCHW tensor is made first); flags = 0.
    From 64 reduction channels on the forward pass runs the f16 implicit GEMM between half transposes instead (CONV_NCHW_HALF_F16)."""
    n, h, w_, c, k, kh, kw, stride, border = case
    if n > 64 and "emu" in request.node.name:
        pytest.skip("528 partial rows (264 images x two 64-pixel tiles: the grouped level of colsum_partials_f16): on the MI355X only, the emulator would take minutes")
    rng = np.random.default_rng(9)
    a = hrnd(rng, n, h, w_, c)
    wt = hrnd(rng, k, kh, kw, c, scale=2.0 / np.sqrt(kh * kw * c))
    bias = hrnd(rng, k)
    hint = nnc.HINT(stride, border)
    oh = (h + 2 * border[0] - kh) // stride[0] + 1
    ow = (w_ + 2 * border[1] - kw) // stride[1] + 1
    nchw = lambda t: np.ascontiguousarray(t.transpose(0, 3, 1, 2))
    s0, n0 = _half_counts(backend)
    fcmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, kh, kw, c)
    res = {}
    names = _kernel_records(backend, lambda: res.update(r=exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [nchw(a), nchw(wt), bias], [np.zeros((n, k, oh, ow), H)], "NCHW")))
    r1, got = res["r"]
    assert any("mfma_gemm_f16" in x for x in names) == (c >= 64), names
    assert _half_counts(backend)[0] == s0  # nothing went through half_stage.cpp's fp32 images
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, hint, 0, [a.astype(F), wt.astype(F), bias.astype(F)], [np.zeros((n, oh, ow, k), F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    _close(got[0], nchw(want[0]))
    g = hrnd(rng, n, oh, ow, k, scale=1.0 / np.sqrt(oh * ow))
    bcmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, kh, kw, c)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, [nchw(g), nchw(a), nchw(wt)], [np.zeros((n, c, h, w_), H), np.zeros((k, c, kh, kw), H), np.zeros(k, H)], "NCHW")
    assert _half_counts(backend)[0] == s0
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, bcmd, hint, 0, [g.astype(F), a.astype(F), wt.astype(F)], [np.zeros((n, h, w_, c), F), np.zeros((k, kh, kw, c), F), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    _close(got[0], nchw(want[0]))
    _close(got[1], nchw(want[1]))
    _close(got[2], want[2])


# ---- rows with native half-precision kernels (half_stage.cpp's table): halves loaded / stored, fp32 arithmetic -----------------
def _half_counts(L):
    """(tensors staged through fp32 images, tensors handed to kernels as halves) so far -- nnc_mi355x_debug_half_counts"""
    import ctypes
    a, b = ctypes.c_long(0), ctypes.c_long(0)
    L.dll.nnc_mi355x_debug_half_counts(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def test_native_half_relu_ewsum_exact(backend, ref_lib):
    rng = np.random.default_rng(21)
    shape = (3, 10, 9, 12)  # 3240 elements: 8-wide body + a tail
    a, b, c = hrnd(rng, *shape), hrnd(rng, *shape), hrnd(rng, *shape)
    s0, n0 = _half_counts(backend)
    got, want = _pair(backend, ref_lib, nnc.CMD_RELU_FORWARD(), nnc.NO_HINT, 0, [a], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0].astype(H))
    assert _half_counts(backend) == (s0, n0 + 2)  # both tensors went to the kernel as halves, no fp32 image was made
    y = got[0]
    g = hrnd(rng, *shape)
    got, want = _pair(backend, ref_lib, nnc.CMD_RELU_BACKWARD(), nnc.NO_HINT, 0, [g, None, y], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0].astype(H))
    got, want = _pair(backend, ref_lib, nnc.CMD_EWSUM_FORWARD(), nnc.NO_HINT, 0, [a, b], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0].astype(H))  # one rounding of the fp32 sum
    got, want = _pair(backend, ref_lib, nnc.CMD_EWSUM_FORWARD(), nnc.NO_HINT, 0, [a, b, c], [np.zeros_like(a)])
    _close(got[0], want[0], tol=1e-3)
    got, want = _pair(backend, ref_lib, nnc.CMD_EWSUM_BACKWARD(), nnc.NO_HINT, 0, [g, a, b, y], [np.zeros_like(a), np.zeros_like(a)])
    assert np.array_equal(got[0], g) and np.array_equal(got[1], g)


@pytest.mark.parametrize("fmt,shape", [("NCHW", (4, 6, 8, 8)), ("NCHW", (3, 5, 7, 7)), ("NHWC", (4, 6, 6, 16)), ("NCHW", (2, 3, 48, 56))], ids=["nchw-vec", "nchw-odd", "nhwc", "nchw-large-planes"])
def test_native_half_batch_norm(backend, ref_lib, fmt, shape):
    """x, y, g, h in CCV_16F and the statistics in fp32 -- what the half-precision trainers issue.  Oracle: the reference's CPU
    batch norm in fp32 on the same half-rounded x / g."""
    from harness import make_tensors
    rng = np.random.default_rng(22)
    caxis = 1 if fmt == "NCHW" else 3
    C = shape[caxis]
    s4 = tuple(C if k == caxis else 1 for k in range(4))
    axes = tuple(k for k in range(4) if k != caxis)
    x = hrnd(rng, *shape, scale=2.0)
    g = hrnd(rng, *shape)
    scale, bias = (rng.random(C, dtype=F) + F(1.0)), (rng.random(C, dtype=F) - F(0.5))
    mean, var = (rng.random(C, dtype=F) - F(0.5)), rng.random(C, dtype=F) + F(0.5)
    cmd = nnc.CMD_BATCH_NORM_FORWARD(1e-4, 0, 0.9, *axes)

    def run(lib, mem, xx, gg, backend_id=None):
        r = lambda a: a.reshape(s4).copy()
        tx, tg = make_tensors(lib, mem, [xx, gg], fmt)
        ts = make_tensors(lib, mem, [r(scale), r(bias), r(mean), r(var)], fmt)
        ty, th = make_tensors(lib, mem, [np.zeros_like(xx), np.zeros_like(xx)], fmt)
        tsm, tsi, tds, tdb = make_tensors(lib, mem, [np.zeros(s4, F) for _ in range(4)], fmt)
        c = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        cb = nnc.CMD_BATCH_NORM_BACKWARD(1e-4, 0, 0.9, *axes)
        if backend_id is not None:
            c.backend = backend_id; cb.backend = backend_id
        assert lib.cmd_exec(c, nnc.NO_HINT, 0, [tx] + ts, [ty, ts[2], ts[3], tsm, tsi]) == 0
        assert lib.cmd_exec(cb, nnc.NO_HINT, 0, [tg] + [None] * 4 + [tx, ts[0]] + [None] * 6 + [tsm, tsi], [th, tds, tdb]) == 0
        return [t.numpy() for t in (ty, th, ts[2], ts[3], tsm, tsi, tds, tdb)]

    s0, n0 = _half_counts(backend)
    got = run(backend, nnc.GPU_MEMORY, x, g)
    assert _half_counts(backend) == (s0, n0 + 5)  # x, y; g, x, h -- the fp32 statistics need no image
    want = run(ref_lib, nnc.CPU_MEMORY, x.astype(F), g.astype(F), nnc.BACKEND_CPU_REF)
    assert got[0].dtype == H and got[1].dtype == H
    _close(got[0], want[0], tol=2e-3)
    _close(got[1], want[1], tol=2e-3)
    for a, b, what in zip(got[2:], want[2:], ("mean", "var", "saved_mean", "saved_inv_std", "dscale", "dbias")):
        np.testing.assert_allclose(a.reshape(-1), b.reshape(-1), rtol=2e-4, atol=2e-5, err_msg=what)


@pytest.mark.parametrize("fmt", ["NCHW", "NHWC"])
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_native_half_pooling(backend, ref_lib, fmt, kind):
    rng = np.random.default_rng(23)
    n, h, w, c = 1, 9, 8, 6  # (the reference's CPU pooling walks one image; batches are covered by the fp32 tests of the same kernels)
    a = hrnd(rng, n, h, w, c)
    hint = nnc.HINT((2, 2), (1, 1))
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    tr = (lambda t: np.ascontiguousarray(t.transpose(0, 3, 1, 2))) if fmt == "NCHW" else (lambda t: t)
    fcmd = nnc.CMD_MAX_POOL_FORWARD(3, 3) if kind == "max" else nnc.CMD_AVERAGE_POOL_FORWARD(3, 3)
    bcmd = nnc.CMD_MAX_POOL_BACKWARD(3, 3) if kind == "max" else nnc.CMD_AVERAGE_POOL_BACKWARD(3, 3)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, fcmd, hint, 0, [tr(a)], [tr(np.zeros((n, oh, ow, c), H))], fmt)
    # the oracle in NHWC (the reference's CPU pooling reads its tensors as NHWC), compared in the backend's layout
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, hint, 0, [a.astype(F)], [np.zeros((n, oh, ow, c), F)], backend=nnc.BACKEND_CPU_REF)
    want = [tr(want[0])]
    assert r1 == 0 and r2 == 0 and got[0].dtype == H
    if kind == "max":
        assert np.array_equal(got[0], want[0].astype(H))
    else:
        _close(got[0], want[0], tol=1e-3)
    y = got[0]
    g = hrnd(rng, n, oh, ow, c)
    ins = [tr(g), tr(a), y] if kind == "max" else [tr(g)]
    untr = (lambda t: np.ascontiguousarray(t.transpose(0, 2, 3, 1))) if fmt == "NCHW" else (lambda t: t)
    r1, gb = exec_on(backend, nnc.GPU_MEMORY, bcmd, hint, 0, ins, [tr(np.zeros_like(a))], fmt)
    r2, wb = exec_on(ref_lib, nnc.CPU_MEMORY, bcmd, hint, 0, [untr(t).astype(F) for t in ins], [np.zeros(a.shape, F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0 and gb[0].dtype == H
    _close(gb[0], tr(wb[0]), tol=1e-3)


@pytest.mark.parametrize("nesterov", [0, 1])
def test_native_half_sgd(backend, ref_lib, nesterov):
    """gradient, parameter and momentum all in CCV_16F (the cifar-10 trainer's half-precision mode): one kernel, fp32 arithmetic."""
    rng = np.random.default_rng(24)
    g, a, m = hrnd(rng, 1003), hrnd(rng, 1003), hrnd(rng, 1003, scale=0.1)
    cmd = nnc.CMD_SGD_FORWARD(nesterov, 0.01, 0.5, 0.0005, 0.9, 0.0 if nesterov else 0.1)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [g, a, m], [np.zeros_like(a), np.zeros_like(m)])
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [g.astype(F), a.astype(F), m.astype(F)], [np.zeros(1003, F), np.zeros(1003, F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    for x, y in zip(got, want):
        assert x.dtype == H
        _close(x, y, tol=1e-3)


@pytest.mark.parametrize("nesterov", [0, 1])
def test_native_half_sgd_in_16_byte_accesses_equals_the_scalar_kernel(backend, ref_lib, nesterov):
    """Tensors of a multiple of eight halves on 16-byte boundaries take sgd_kernel_h8 (eight halves per lane); the same numbers one element further into the
    same allocations are not aligned and take the scalar kernel: bit-identical updates, and both within tolerance of the reference's fp32 update."""
    from harness import make_tensors
    lib = backend
    rng = np.random.default_rng(25)
    n = 4104
    g, a, m = hrnd(rng, n), hrnd(rng, n), hrnd(rng, n, scale=0.1)
    cmd = nnc.CMD_SGD_FORWARD(nesterov, 0.01, 0.5, 0.0005, 0.9, 0.0 if nesterov else 0.1)
    outs = {}
    for off in (0, 1):
        pad = lambda x: np.concatenate([np.zeros(off, H), x, np.zeros(8 - off, H)])
        tg, ta, tm, tb, tn = make_tensors(lib, nnc.GPU_MEMORY, [pad(g), pad(a), pad(m), np.zeros(n + 8, H), np.zeros(n + 8, H)])
        ins = [t.alias((n,), off) for t in (tg, ta, tm)]
        ots = [t.alias((n,), off) for t in (tb, tn)]
        assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, ins, ots) == 0
        lib.stream_wait(None)
        full = [tb.numpy(), tn.numpy()]
        for f in full:  # nothing outside the aliased range was written
            assert (f[:off] == 0).all() and (f[off + n:] == 0).all()
        outs[off] = [f[off:off + n] for f in full]
    for x, y in zip(outs[0], outs[1]):
        assert (x.view(np.uint16) == y.view(np.uint16)).all()
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [g.astype(F), a.astype(F), m.astype(F)], [np.zeros(n, F), np.zeros(n, F)], backend=nnc.BACKEND_CPU_REF)
    assert r2 == 0
    for x, y in zip(outs[0], want):
        _close(x, y, tol=1e-3)


# ---- the half-precision contraction kernel's two staging widths (mfma_gemm_f16.h: TileFetchH, 8-byte chunks of four halves; TileFetchH8, 16-byte chunks of eight where
# channel counts and strides are multiples of eight): the same LDS images, the same fragments -- bit-identical results
@pytest.mark.parametrize("what", ["conv-nhwc", "conv-nchw", "conv-nchw-s2", "gemm"])
def test_half_contraction_chunks_of_eight_equal_chunks_of_four(backend, what):
    rng = np.random.default_rng(17)
    if what == "gemm":
        runs = [(nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [hrnd(rng, 72, 136), hrnd(rng, 80, 136), hrnd(rng, 80)], [np.zeros((72, 80), H)], "NHWC"),
                (nnc.CMD_GEMM_BACKWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [hrnd(rng, 72, 80), hrnd(rng, 72, 136), hrnd(rng, 80, 136)], [np.zeros((72, 136), H), np.zeros((80, 136), H), np.zeros(80, H)], "NHWC")]
    else:
        fmt = "NHWC" if what == "conv-nhwc" else "NCHW"
        n, h, w, c, k = 3, 10, 12, 64, 72
        stride = (2, 2) if what.endswith("s2") else (1, 1)
        hint = nnc.HINT(stride, (1, 1))
        oh, ow = (h + 2 - 3) // stride[0] + 1, (w + 2 - 3) // stride[1] + 1
        sh = (lambda *d: d) if fmt == "NHWC" else (lambda nn, hh, ww, cc: (nn, cc, hh, ww))
        a, wt, g = hrnd(rng, *sh(n, h, w, c)), hrnd(rng, *sh(k, 3, 3, c), scale=0.05), hrnd(rng, *sh(n, oh, ow, k), scale=0.1)
        runs = [(nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), hint, [a, wt, hrnd(rng, k)], [np.zeros(sh(n, oh, ow, k), H)], fmt),
                (nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c), hint, [g, a, wt], [np.zeros(sh(n, h, w, c), H), np.zeros(sh(k, 3, 3, c), H), np.zeros(k, H)], fmt)]
    for cmd, hint, ins, outs, fmt in runs:
        res = {}
        try:
            for mode in (1, 0):
                backend.tune_set("GEMM_HALF_CHUNK8", mode)
                r, got = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, ins, outs, fmt)
                assert r == 0
                res[mode] = got
        finally:
            backend.tune_set("GEMM_HALF_CHUNK8", 1)
        for x, y in zip(res[1], res[0]):
            assert np.array_equal(x.view(np.uint16), y.view(np.uint16))


@pytest.mark.parametrize("shape", [(2, 64, 8, 8), (3, 72, 8, 16), (1, 128, 16, 16), (2, 8, 4, 2), (2, 200, 28, 28), (2, 12, 14, 14)])
def test_half_layout_transposes_in_16_byte_accesses(backend, shape):
    """Round 6: NCHW <-> NHWC of CCV_16F tensors (FORMAT_TRANSFORM, and the re-layouts around the f16 3 x 3 convolutions) through transpose_half8_kernel -- a 64 x 64
    tile of halves in LDS, read back through the LDS transpose read, 16 bytes per lane both ways (ccv_amd/csrc/cmd_util.cpp).  Exact (a permutation of bits); sizes
    with ragged tiles, and one (14 x 14 planes) that takes the 8-byte kernel.  lib/nnc/cmd/util/ccv_nnc_util_cpu_ref.c:996-1082 is the spec."""
    L = backend
    n, c, h, w = shape
    x = np.random.default_rng(9).standard_normal((n, c, h, w)).astype(np.float16)
    a = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, nnc.CCV_16F, (n, c, h, w), 0), x)
    b = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_16F, (n, h, w, c), 0), np.zeros((n, h, w, c), np.float16))
    assert L.cmd_exec(nnc.CMD_FORMAT_TRANSFORM_FORWARD(), nnc.NO_HINT, 0, [a], [b]) == 0
    assert np.array_equal(b.numpy(), x.transpose(0, 2, 3, 1))
    a2 = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, nnc.CCV_16F, (n, c, h, w), 0), np.zeros((n, c, h, w), np.float16))
    assert L.cmd_exec(nnc.CMD_FORMAT_TRANSFORM_FORWARD(), nnc.NO_HINT, 0, [b], [a2]) == 0
    assert np.array_equal(a2.numpy(), x)


@pytest.mark.parametrize("label_kind", ["f32", "i32", "dense", "half"])
@pytest.mark.parametrize("trim", [(0.0, 1.0), (0.1, 0.9)])
def test_native_half_softmax_crossentropy(backend, ref_lib, label_kind, trim):
    """Round 6: SOFTMAX_CROSSENTROPY forward / backward keep the logits, the softmax and the gradient as CCV_16F in their own memory (cmd_loss.cpp templated on the
    element type; g_native_half) -- the last rows of the trainers' steps that went through fp32 images.  The arithmetic is the fp32 kernel's with ONE rounding per
    stored value, so the result EQUALS the reference's fp32 result on the same half inputs rounded to half; the loss (fp32 per row, or half when the host's loss
    tensor is half) and the labels keep their small images."""
    rng = np.random.default_rng(31)
    n, c = 7, 45
    a = hrnd(rng, n, c, scale=3)
    idx = rng.integers(0, c, n)
    if label_kind == "f32":
        label = idx.astype(F)
    elif label_kind == "i32":
        label = idx.astype(np.int32)
    elif label_kind == "half":
        label = idx.astype(H)  # the half trainers' label tensors (exact up to 2048 classes)
    else:
        label = rng.random((n, c), dtype=F)
        label /= label.sum(1, keepdims=True)
    s0, n0 = _half_counts(backend)
    got, want = _pair(backend, ref_lib, nnc.CMD_SOFTMAX_CROSSENTROPY_FORWARD(*trim), nnc.NO_HINT, 0, [a, label], [np.zeros(n, F), np.zeros((n, c), H)])
    s1, n1 = _half_counts(backend)
    assert n1 == n0 + 2 and s1 - s0 == (1 if label_kind == "half" else 0)  # logits and softmax as halves; only a half label tensor gets an image
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-5)
    assert got[1].dtype == H and np.array_equal(got[1], want[1].astype(H))
    g = ((rng.random(n, dtype=F) - 0.5) * 2).astype(F)
    d = got[1]
    got, want = _pair(backend, ref_lib, nnc.CMD_SOFTMAX_CROSSENTROPY_BACKWARD(*trim), nnc.NO_HINT, 0, [g, None, None, label, None, d], [np.zeros((n, c), H)])
    assert _half_counts(backend)[1] == n1 + 2
    assert got[0].dtype == H
    # (the reference's label-smoothing backward multiplies in another order: one half ulp at most)
    np.testing.assert_allclose(got[0].astype(F), want[0].astype(H).astype(F), rtol=1e-3, atol=1e-6)
    if trim == (0.0, 1.0):
        assert np.array_equal(got[0], want[0].astype(H))
