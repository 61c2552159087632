"""The BENCHMARKED configuration against the oracle: every distinct VGG-D convolution layer at its
FULL spatial size, BASELINE config 1 exactly, and one whole VGG-D training step -- HIP path vs the reference's own CPU
backend (oracle/_ref/libccv_ref.so, CPU_REF direct loops) on the same seeded inputs.

Inputs follow the reference's GPU-vs-CPU tests (test/int/nnc/cudnn.tests.c:37-45): a ~ U[0,1), w ~ U[0,1) / (C*kh*kw),
bias[i] = i / K.  Tolerances, written where they are applied:
  * forward (all terms positive, no cancellation): ELEMENTWISE 1e-4 relative -- the north star's bound -- and the reference
    test's own absolute 1e-4 (test/int/nnc/cudnn.tests.c:84, REQUIRE_ARRAY_EQ_WITH_TOLERANCE is an absolute bound, test/case.h:145)
  * gradients (signed g: the sums cancel, an elementwise relative bound is meaningless at the zero crossings):
    |got - want| <= 1e-4 * max|want| -- 1e-4 of the tensor's scale.
Every conv algorithm the backend offers (cmd.algorithm = 0 implicit GEMM, 1 Winograd via HBM, 2 fused Winograd where the
geometry allows, -1 the backend's own choice) is held to the same bound.
"""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_on

F = np.float32

# (name, H = W of the layer input, C, K, border): the nine distinct conv geometries of vgg_d_params (bin/vgg_models.inc:361-838)
VGG_D_CONVS = [
    ("conv1_1", 225, 3, 64, 0),
    ("conv1_2", 223, 64, 64, 1),
    ("conv2_1", 111, 64, 128, 1),
    ("conv2_2", 111, 128, 128, 1),
    ("conv3_1", 55, 128, 256, 1),
    ("conv3_2", 55, 256, 256, 1),
    ("conv4_1", 27, 256, 512, 1),
    ("conv4_2", 27, 512, 512, 1),
    ("conv5_1", 13, 512, 512, 1),
]


def _ref_inputs(rng, n, h, c, k):
    a = rng.random((n, h, h, c), dtype=F)
    w = (rng.random((k, 3, 3, c), dtype=F) / F(c * 9)).astype(F)
    bias = (np.arange(k, dtype=F) / F(k)).astype(F)
    return a, w, bias


def _algos(lib, fwd):
    n = [r for name, c, b, r in lib.registry() if c == nnc.CMD["CONVOLUTION_FORWARD" if fwd else "CONVOLUTION_BACKWARD"]][0].algorithms
    return [-1] + list(range(n))


def _scale_close(got, want, what):
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    scale = float(np.abs(want).max())
    assert err <= 1e-4 * scale, "%s: max |diff| %.3g vs 1e-4 x scale %.3g" % (what, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("layer", VGG_D_CONVS, ids=[l[0] for l in VGG_D_CONVS])
def test_vgg_d_conv_layer_full_size_vs_cpu_ref(gpu_lib, ref_lib, layer):
    name, h, c, k, border = layer
    n = 2
    rng = np.random.default_rng(100 + [l[0] for l in VGG_D_CONVS].index(name))
    a, w, bias = _ref_inputs(rng, n, h, c, k)
    oh = h + 2 * border - 2
    hint = nnc.HINT((1, 1), (border, border))
    fwd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    r, want = exec_on(ref_lib, nnc.CPU_MEMORY, fwd, hint, 0, [a, w, bias], [np.zeros((n, oh, oh, k), F)], backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    for algo in _algos(gpu_lib, True):
        fwd.algorithm = algo
        r, got = exec_on(gpu_lib, nnc.GPU_MEMORY, fwd, hint, 0, [a, w, bias], [np.full((n, oh, oh, k), 7, F)])
        assert r == 0, (name, algo, r)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=0, err_msg="%s forward algorithm %d" % (name, algo))  # north star: 1e-4 relative, elementwise
        assert float(np.abs(got[0] - want[0]).max()) <= 1e-4  # the reference test's own absolute bound
    g = ((rng.random((n, oh, oh, k), dtype=F) - F(0.5)) * F(2)).astype(F)
    bwd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    r, want = exec_on(ref_lib, nnc.CPU_MEMORY, bwd, hint, 0, [g, a, w], [np.zeros_like(a), np.zeros_like(w), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    for algo in _algos(gpu_lib, False):
        bwd.algorithm = algo
        r, got = exec_on(gpu_lib, nnc.GPU_MEMORY, bwd, hint, 0, [g, a, w], [np.full_like(a, 3), np.full_like(w, 5), np.full(k, 9, F)])
        assert r == 0, (name, algo, r)
        for i, what in enumerate(("dgrad", "wgrad", "dbias")):
            _scale_close(got[i], want[i], "%s %s algorithm %d" % (name, what, algo))


@pytest.mark.gpu
def test_conv1_2_wgrad_reduction_over_802816_tiles(gpu_lib, ref_lib):
    """The numerically worst contraction of the benchmark: conv1_2's filter gradient at batch 256 sums over 256 x 223 x 223
    pixels (802 816 Winograd tiles).  The CPU oracle cannot walk 256 full-size images inside a test, so the batch is built
    from TWO base images with per-image factors: a_i = alpha_i * A[i % 2], g_i = beta_i * G[i % 2]  =>  (bilinearity)
      dw = sum_b (sum_{i % 2 == b} alpha_i beta_i) * dw(A[b], G[b]),   dbias = sum_b (sum beta_i) * dbias(G[b]),
    with the per-base gradients from CPU_REF and the factor sums in float64.  Signed factors: the 128 terms per base cancel
    as a real batch's do.  Bound: 1e-4 of the tensor's scale, every algorithm."""
    n, h, c, k = 256, 223, 64, 64
    rng = np.random.default_rng(11)
    A, w, _ = _ref_inputs(rng, 2, h, c, k)
    G = ((rng.random((2, h, h, k), dtype=F) - F(0.5)) * F(2)).astype(F)
    alpha = ((rng.random(n) + 0.5) * np.where(rng.random(n) < 0.5, -1, 1)).astype(F)
    beta = (rng.random(n) + 0.5).astype(F)
    hint = nnc.HINT((1, 1), (1, 1))
    bwd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    want_dw = np.zeros(w.shape, np.float64)
    want_db = np.zeros(k, np.float64)
    for b in range(2):
        r, o = exec_on(ref_lib, nnc.CPU_MEMORY, bwd, hint, 0, [G[b:b + 1], A[b:b + 1], w], [None, np.zeros_like(w), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
        assert r == 0
        sel = np.arange(n) % 2 == b
        want_dw += float(np.sum(alpha[sel].astype(np.float64) * beta[sel].astype(np.float64))) * o[1].astype(np.float64)
        want_db += float(np.sum(beta[sel].astype(np.float64))) * o[2].astype(np.float64)
    a = np.empty((n, h, h, c), F)
    g = np.empty((n, h, h, k), F)
    for i in range(n):
        np.multiply(A[i % 2], alpha[i], out=a[i])
        np.multiply(G[i % 2], beta[i], out=g[i])
    ta = gpu_lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, a.shape), a)
    tg = gpu_lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, g.shape), g)
    tw = gpu_lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, w.shape), w)
    del a, g
    for algo in _algos(gpu_lib, False):
        bwd.algorithm = algo
        tdw = gpu_lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, w.shape), np.full_like(w, 5))
        tdb = gpu_lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, (k,)), np.full(k, 9, F))
        assert gpu_lib.cmd_exec(bwd, hint, 0, [tg, ta, tw], [None, tdw, tdb]) == 0
        gpu_lib.stream_wait(None)
        _scale_close(tdw.numpy(), want_dw, "conv1_2 wgrad at batch 256, algorithm %d" % algo)
        _scale_close(tdb.numpy(), want_db, "conv1_2 dbias at batch 256, algorithm %d" % algo)


@pytest.mark.parametrize("fmt", ["NHWC", "NCHW"])
def test_baseline_config_1(backend, ref_lib, fmt):
    """BASELINE.json configs[0]: 3x3 fp32 CCV_NNC_CONVOLUTION_FORWARD on 1 x 3 x 224 x 224 -> 64 channels, border 1, against
    the reference's CPU_REF backend (the configuration the reference itself can run without a GPU), in both layouts the
    backend registers.  Elementwise 1e-4 relative (all terms positive) and the reference tests' absolute 1e-4."""
    rng = np.random.default_rng(3)
    a, w, bias = _ref_inputs(rng, 1, 224, 3, 64)
    hint = nnc.HINT((1, 1), (1, 1))
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, 64, 3, 3, 3)
    if fmt == "NCHW":
        a_in, w_in, out0 = np.ascontiguousarray(a.transpose(0, 3, 1, 2)), np.ascontiguousarray(w.transpose(0, 3, 1, 2)), np.zeros((1, 64, 224, 224), F)
    else:
        a_in, w_in, out0 = a, w, np.zeros((1, 224, 224, 64), F)
    # the CPU_REF convolution reads NHWC and NCHW alike through tensor views (conv_cpu_ref.c:13-106)
    r, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a_in, w_in, bias], [out0], fmt=fmt, backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    r, got = exec_on(backend, nnc.GPU_MEMORY, cmd, hint, 0, [a_in, w_in, bias], [out0 + 7], fmt=fmt)
    assert r == 0
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=0)
    assert float(np.abs(got[0] - want[0]).max()) <= 1e-4
    if fmt == "NCHW":  # and the two layouts agree with each other on the oracle side (guards the test's own transposes)
        r, want2 = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, w, bias], [np.zeros((1, 224, 224, 64), F)], backend=nnc.BACKEND_CPU_REF)
        np.testing.assert_allclose(want[0], want2[0].transpose(0, 3, 1, 2), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_vgg_d_full_step_n2_vs_cpu_ref(gpu_lib, ref_lib):
    """One whole VGG-D training step (forward + backward + SGD, 225 x 225 x 3 input, the benchmark's command sequence and
    its default algorithm choices) at batch 2 against the reference's CPU backend driven through the same commands.
      forward : per-image loss and softmax, and EVERY activation (post-ReLU conv / fc outputs, pool outputs) to 1e-4 of the
                tensor's scale;
      backward: EVERY parameter gradient to 1e-4 of its scale and EVERY updated parameter -- with the oracle's backward pass
                reading the SAME forward state as ours (the GPU's activations are copied into the oracle's tensors first).
    Why the hand-over: ReLU and max-pool route gradients by comparisons on the forward values.  Two fp32 forward passes that
    agree to 1e-5 still disagree on the sign / the arg-max of a few activations in a few million, and every such flip moves a
    whole gradient element: run end to end, the two backward passes differ by ~sqrt(flipped fraction) ~ 0.5 % in L2 (measured
    on the MI355X: 0.9 % at conv1_1 falling to 0.05 % at conv5_3; the loss agrees to 3e-6) -- a property of comparing ANY two
    fp32 implementations through 16 ReLU layers, not of either one.  The count of disagreeing ReLU masks is reported."""
    from ccv_amd.vgg import VGGD
    rng = np.random.default_rng(17)
    x = rng.random((2, 225, 225, 3), dtype=F)
    y = rng.integers(0, 1000, 2)
    net = VGGD(gpu_lib, 2, seed=0)
    net.set_input(x, y)
    gpu_lib.stream_wait(None)
    net.forward()
    gpu_lib.stream_wait(None)
    from oracle_vgg import OracleVGGD
    ref = OracleVGGD(ref_lib, 2, memory=nnc.CPU_MEMORY, seed=0, backend=nnc.BACKEND_CPU_REF)
    ref.set_input(x, y)
    ref.forward()
    np.testing.assert_allclose(net.loss.numpy(), ref.loss.numpy(), rtol=1e-4, atol=0)
    np.testing.assert_allclose(net.softmax.numpy(), ref.softmax.numpy(), rtol=1e-3, atol=1e-7)
    bad, flips, total = [], 0, 0
    for i, (n, m) in enumerate(zip(net.nodes, ref.nodes)):
        got, want = n["b"].numpy(), m["b"].numpy()
        err, scale = float(np.abs(got - want).max()), float(np.abs(want).max())
        if not err <= 1e-4 * scale:
            bad.append("activation of node %d (%s) %s: max |diff| %.3g vs scale %.3g" % (i, n["kind"], got.shape, err, scale))
        if n.get("relu"):
            flips += int(np.count_nonzero((got > 0) != (want > 0)))
            total += got.size
        m["b"].array[...] = got  # hand the forward state over: both backward passes now route through identical masks
    ref.softmax.array[...] = net.softmax.numpy()
    print("ReLU masks that differ between the two forward passes: %d of %d" % (flips, total))
    net.backward()
    net.update()
    gpu_lib.stream_wait(None)
    ref.backward()
    ref.update()
    for i, ((p, d, _), (q, e, _)) in enumerate(zip(net.params, ref.params)):
        dg, de = d.numpy().astype(np.float64), e.numpy().astype(np.float64)
        err, scale = float(np.abs(dg - de).max()), float(np.abs(de).max())
        if not err <= 1e-4 * scale:  # 1e-4 of the gradient tensor's scale
            bad.append("gradient %d %s: max |diff| %.3g vs scale %.3g (relative L2 %.3g)" % (i, d.dims, err, scale, float(np.linalg.norm(dg - de) / np.linalg.norm(de))))
        pg, pe = p.numpy(), q.numpy()
        if not np.allclose(pg, pe, rtol=1e-5, atol=1e-7):
            bad.append("updated parameter %d %s: max |diff| %.3g" % (i, p.dims, float(np.abs(pg - pe).max())))
    assert not bad, "\n".join(bad)


WINOGRAD_UNIT_CASES = [
    # the reference's own Winograd-vs-direct unit tests, test/unit/nnc/winograd.tests.c:14-134: (name, h = w, C, K, bias)
    ("56x56 non-uniform weights", 56, 128, 128, True),
    ("55x55 non-uniform weights", 55, 128, 128, True),
    ("224x224 RGB", 224, 3, 128, True),
    ("56x56 no bias", 56, 128, 128, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WINOGRAD_UNIT_CASES, ids=[c[0] for c in WINOGRAD_UNIT_CASES])
def test_reference_winograd_unit_shapes_on_the_hip_path(gpu_lib, ref_lib, case):
    """the reference pins its CPU Winograd (CPU_OPT, algorithm 2) to its direct loops (CPU_REF) under
    REQUIRE_TENSOR_EQ on these shapes -- 3-d H x W x C tensors, border 1, weights u / (9 C), bias i / K (winograd.tests.c:14-134).
    Replayed here with the HIP path in the middle: every algorithm of this backend (implicit GEMM, Winograd via HBM, fused
    Winograd, its own choice) against CPU_REF at the north star's 1e-4 relative and the reference conv tests' absolute 1e-4,
    and CPU_OPT's Winograd against CPU_REF as the reference's own test has it (tensor_eq = REQUIRE_TENSOR_EQ semantics)."""
    from harness import tensor_eq
    name, h, c, k, with_bias = case
    rng = np.random.default_rng(300 + h + c)
    a = rng.random((h, h, c), dtype=F)
    w = (rng.random((k, 3, 3, c), dtype=F) / F(9 * c)).astype(F)
    bias = (np.arange(k, dtype=F) / F(k)).astype(F) if with_bias else np.zeros(k, F)
    hint = nnc.HINT((1, 1), (1, 1))
    fwd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    r, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, fwd, hint, 0, [a, w, bias], [np.zeros((h, h, k), F)], backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    opt = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    opt.algorithm = 2  # CCV_NNC_CMD_OPT_CONV_ALGO_WINOGRAD
    r, (cpu_wino,) = exec_on(ref_lib, nnc.CPU_MEMORY, opt, hint, 0, [a, w, bias], [np.zeros((h, h, k), F)], backend=nnc.BACKEND_CPU_OPT)
    assert r == 0 and tensor_eq(cpu_wino, want), "the reference's own CPU Winograd vs its direct loops"
    for algo in _algos(gpu_lib, True):
        fwd.algorithm = algo
        r, (got,) = exec_on(gpu_lib, nnc.GPU_MEMORY, fwd, hint, 0, [a, w, bias], [np.full((h, h, k), 7, F)])
        assert r == 0, (name, algo, r)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=0, err_msg="%s algorithm %d" % (name, algo))
        assert float(np.abs(got - want).max()) <= 1e-4


# ResNet-50 v1d's stride-2 convolutions (bin/nnc/imagenet.c:17-98: the 3 x 3 of the first block of stages 2-4, the stem) at their
# full spatial sizes: the parity-class data gradient (cmd_conv.cpp: conv_dgrad_parity) against the reference's CPU backend.
RESNET_STRIDE2 = [
    ("stage2", 56, 128, 128, 3, 1),
    ("stage3", 28, 256, 256, 3, 1),
    ("stage4", 14, 512, 512, 3, 1),
    ("stem-7x7-class", 112, 16, 32, 7, 3),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", RESNET_STRIDE2, ids=[c[0] for c in RESNET_STRIDE2])
def test_resnet_stride2_data_gradient_full_size_vs_cpu_ref(gpu_lib, ref_lib, case):
    _, hw, c, k, ks, pad = case
    rng = np.random.default_rng(31)
    n = 1
    hint = nnc.HINT((2, 2), (pad, pad))
    oh = (hw + 2 * pad - ks) // 2 + 1
    a = rng.random((n, hw, hw, c), dtype=F)
    w = ((rng.random((k, ks, ks, c), dtype=F) - F(0.5)) / F(c * ks * ks)).astype(F)
    g = (rng.random((n, oh, oh, k), dtype=F) - F(0.5))
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, ks, ks, c)
    outs = [np.zeros_like(a), np.zeros_like(w), np.zeros(k, F)]
    r1, got = exec_on(gpu_lib, nnc.GPU_MEMORY, cmd, hint, 0, [g, a, w], outs)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [g, a, w], outs, backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    for x, y, what in zip(got, want, ("h", "dw", "dbias")):
        bound = 1e-4 * float(np.abs(y).max())  # 1e-4 of the tensor's scale (signed sums: see the module docstring)
        assert float(np.abs(x - y).max()) <= bound, what
