"""Parity of the activation / softmax / Adam / AdamW / RMSProp rows (ccv_amd/csrc/cmd_act_opt.cpp) with the reference's CPU
backend on identical inputs.  The reference promotes to double inside the math calls and stores fp32; the kernels compute in
fp32: a few ulp (1e-6 relative) is the tolerance, exact for the piecewise-linear ones."""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_pair, exec_on

F = np.float32
SHAPES = [(7,), (4, 5, 6, 3), (3, 1027)]


def _x(shape, seed=0, scale=3.0):
    return ((np.random.default_rng(seed).random(shape, dtype=F) - 0.5) * 2 * scale).astype(F)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", ["SIGMOID", "TANH", "SWISH", "GELU", "GELU_TANH", "LEAKY_RELU"])
def test_activation_forward_backward(backend, ref_lib, name, shape):
    a, g = _x(shape, 1), _x(shape, 2, 1.0)
    if name.startswith("GELU"):
        fwd, bwd = nnc.CMD_GELU_FORWARD(int(name.endswith("TANH"))), nnc.CMD_GELU_BACKWARD(int(name.endswith("TANH")))
    elif name == "LEAKY_RELU":
        fwd, bwd = nnc.CMD_LEAKY_RELU_FORWARD(0.2), nnc.CMD_LEAKY_RELU_BACKWARD(0.2)
    else:
        fwd, bwd = nnc.generic_cmd(name + "_FORWARD"), nnc.generic_cmd(name + "_BACKWARD")
    got, want = exec_pair(backend, ref_lib, fwd, nnc.NO_HINT, 0, [a], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-6, atol=1e-7)
    b = want[0]
    from_output = name in ("SIGMOID", "TANH", "LEAKY_RELU")
    ins = [g, None, b] if from_output else [g, a, None]  # swish_cpu_ref.c:36 asserts input_size == 3
    got, want = exec_pair(backend, ref_lib, bwd, nnc.NO_HINT, 0, ins, [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=4e-6, atol=2e-7)
    if name in ("SIGMOID", "TANH"):
        # no incoming gradient = ones (sigmoid_cpu_ref.c:57-62 has the branch, but its shape loop dereferences g first and
        # crashes on NULL, so the expected value is the formula itself)
        from harness import exec_on
        r, got = exec_on(backend, nnc.GPU_MEMORY, bwd, nnc.NO_HINT, 0, [None, None, b], [np.zeros_like(a)])
        assert r == 0
        np.testing.assert_allclose(got[0], b * (1 - b) if name == "SIGMOID" else 1 - b * b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(10,), (5, 1000), (3, 7, 11)])
def test_softmax_forward_backward(backend, ref_lib, shape):
    a, g = _x(shape, 3, 4.0), _x(shape, 4, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd("SOFTMAX_FORWARD"), nnc.NO_HINT, 0, [a], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-8)
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd("SOFTMAX_BACKWARD"), nnc.NO_HINT, 0, [g, None, want[0]], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("shape", [(10,), (64, 3, 3, 3), (5, 7)])
@pytest.mark.parametrize("kind", ["adam", "adam_ams", "adamw", "adamw_ams"])
def test_adam(backend, ref_lib, kind, shape):
    rng = np.random.default_rng(7)
    g, a, m = _x(shape, 5, 1.0), _x(shape, 6, 1.0), _x(shape, 7, 0.1)
    v, vm = rng.random(shape, dtype=F) * 0.01, rng.random(shape, dtype=F) * 0.02
    ams = kind.endswith("ams")
    cmd = nnc.CMD_ADAM_FORWARD(3, 0.002, 0.9, 0.98, 0.01, 1e-8, amsgrad=int(ams), scale=0.5, decoupled=kind.startswith("adamw"))
    ins = [g, a, m, v] + ([vm] if ams else [])
    outs = [np.zeros_like(a) for _ in range(4 if ams else 3)]
    got, want = exec_pair(backend, ref_lib, cmd, nnc.NO_HINT, 0, ins, outs)
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=2e-6, atol=1e-8)


@pytest.mark.parametrize("shape", [(10,), (64, 3, 3, 3)])
def test_rmsprop(backend, ref_lib, shape):
    g, a, m = _x(shape, 8, 1.0), _x(shape, 9, 1.0), _x(shape, 10, 0.1)
    v = np.random.default_rng(11).random(shape, dtype=F) * 0.01
    cmd = nnc.CMD_RMSPROP_FORWARD(0.001, 0.0005, 0.9, 0.9, 1e-4, scale=0.5)
    got, want = exec_pair(backend, ref_lib, cmd, nnc.NO_HINT, 0, [g, a, m, v], [np.zeros_like(a) for _ in range(3)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=2e-6, atol=1e-8)


# ---- row losses (ccv_amd/csrc/cmd_loss2.cpp) -----------------------------------------------------------------------------------
LOSS_SHAPES = [(6, 10), (3, 1000), (17,)]


def _rows(shape):
    return (shape[0],) if len(shape) > 1 else (1,)


@pytest.mark.parametrize("shape", LOSS_SHAPES)
@pytest.mark.parametrize("reduce_op", [0, 1])
def test_mse(backend, ref_lib, shape, reduce_op):
    a, b, g = _x(shape, 1, 1.0), _x(shape, 2, 1.0), _x(_rows(shape), 3, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_MSE_FORWARD(reduce_op), nnc.NO_HINT, 0, [a, b], [np.zeros(_rows(shape), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-7)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_MSE_BACKWARD(reduce_op), nnc.NO_HINT, 0, [g, a, b], [np.zeros_like(a), np.zeros_like(a)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-7)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_MSE_BACKWARD(reduce_op), nnc.NO_HINT, 0, [None, a, b], [np.zeros_like(a)])  # no incoming gradient, ha only
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("shape", LOSS_SHAPES)
@pytest.mark.parametrize("beta", [0.5, 300.0])   # rows on either side of the L1-sum threshold
def test_smooth_l1(backend, ref_lib, shape, beta):
    a, b, g = _x(shape, 4, 1.0), _x(shape, 5, 1.0), _x(_rows(shape), 6, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SMOOTH_L1_FORWARD(beta), nnc.NO_HINT, 0, [a, b], [np.zeros(_rows(shape), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-7)
    c = want[0]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_SMOOTH_L1_BACKWARD(beta), nnc.NO_HINT, 0, [g, a, b, c], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("shape", LOSS_SHAPES)
@pytest.mark.parametrize("pos_weight", [1.0, 2.5])
def test_binary_crossentropy(backend, ref_lib, shape, pos_weight):
    rng = np.random.default_rng(7)
    a = (rng.random(shape, dtype=F) * 0.98 + 0.01).astype(F)
    b = (rng.random(shape) < 0.4).astype(F)
    g = _x(_rows(shape), 8, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_BINARY_CROSSENTROPY_FORWARD(pos_weight), nnc.NO_HINT, 0, [a, b], [np.zeros(_rows(shape), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_BINARY_CROSSENTROPY_BACKWARD(pos_weight), nnc.NO_HINT, 0, [g, a, b], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("label_kind", ["f32", "i32", "dense"])
@pytest.mark.parametrize("trim", [(0.0, 1.0), (0.1, 0.9)])
def test_categorical_crossentropy(backend, ref_lib, label_kind, trim):
    rng = np.random.default_rng(9)
    n, c = 6, 50
    a = rng.random((n, c), dtype=F) + 0.05
    a = (a / a.sum(axis=1, keepdims=True)).astype(F)
    idx = rng.integers(0, c, n)
    if label_kind == "dense":
        if trim != (0.0, 1.0):
            pytest.skip("dense labels carry their own smoothing")
        label = rng.random((n, c), dtype=F)
        label = (label / label.sum(axis=1, keepdims=True)).astype(F)
    else:
        label = idx.astype(F if label_kind == "f32" else np.int32)
    g = _x((n,), 10, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_CATEGORICAL_CROSSENTROPY_FORWARD(*trim), nnc.NO_HINT, 0, [a, label], [np.zeros((n,), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)
    got, want = exec_pair(backend, ref_lib, nnc.CMD_CATEGORICAL_CROSSENTROPY_BACKWARD(*trim), nnc.NO_HINT, 0, [g, a, label], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("shape", LOSS_SHAPES)
@pytest.mark.parametrize("pos_weight", [1.0, 2.5])
def test_sigmoid_binary_crossentropy(backend, ref_lib, shape, pos_weight):
    rng = np.random.default_rng(12)
    a, b, g = _x(shape, 13, 3.0), (rng.random(shape) < 0.4).astype(F), _x(_rows(shape), 14, 1.0)
    fwd, bwd = nnc._f1("SIGMOID_BINARY_CROSSENTROPY_FORWARD", pos_weight), nnc._f1("SIGMOID_BINARY_CROSSENTROPY_BACKWARD", pos_weight)
    got, want = exec_pair(backend, ref_lib, fwd, nnc.NO_HINT, 0, [a, b], [np.zeros(_rows(shape), F), np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got[1], want[1], rtol=2e-6, atol=1e-7)
    got, want = exec_pair(backend, ref_lib, fwd, nnc.NO_HINT, 0, [a, b], [None, np.zeros_like(a)])   # "no loss": the sigmoid only
    np.testing.assert_allclose(got[1], want[1], rtol=2e-6, atol=1e-7)
    d = want[1]
    got, want = exec_pair(backend, ref_lib, bwd, nnc.NO_HINT, 0, [g, None, None, b, None, d], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=2e-5, atol=1e-6)


# ---- reduce norm2, element-wise min / max, argmax / argmin (ccv_amd/csrc/cmd_bcast.cpp) -----------------------------------------
def _reduce_cmd(name, *axis):
    c = nnc.generic_cmd(name)
    for i, a in enumerate(axis):
        c.info.reduce.axis[i] = a
    c.info.reduce.count = len(axis)
    return c


@pytest.mark.parametrize("name", ["REDUCE_SUM_FORWARD", "REDUCE_MEAN_FORWARD", "REDUCE_NORM2_FORWARD"])  # (REDUCE_MAX / MIN have no GPU row in the host's table, cmd_bcast.cpp)
@pytest.mark.parametrize("shape,axis", [((3, 5000), (1,)), ((70, 90, 2), (0, 1)), ((9000,), (0,))])
def test_large_reductions_take_the_two_stage_path(backend, ref_lib, name, shape, axis):
    """>= 4096 reduced elements per output: workgroups fold slices of the reduced sub-space, a second kernel folds the slices (cmd_bcast.cpp;
    the serial one-lane-per-output kernel keeps the small cases in the reference's order).  Sums to 1e-5 of the sum of magnitudes."""
    a = _x(shape, 23, 2.0)
    oshape = tuple(1 if i in axis else d for i, d in enumerate(shape))
    got, want = exec_pair(backend, ref_lib, _reduce_cmd(name, *axis), nnc.NO_HINT, 0, [a], [np.zeros(oshape, F)])
    scale = float(np.abs(a).sum(axis=axis).max()) if "NORM2" not in name else float(np.sqrt((a.astype(np.float64) ** 2).sum(axis=axis)).max())
    if "MEAN" in name:
        scale /= np.prod([shape[i] for i in axis])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("shape,axis", [((4, 5, 6), (1,)), ((3, 7), (0,)), ((2, 3, 4, 5), (0, 2)), ((6, 9), (0, 1))])
def test_reduce_norm2(backend, ref_lib, shape, axis):
    a = _x(shape, 21, 2.0)
    oshape = tuple(1 if i in axis else d for i, d in enumerate(shape))
    g = _x(oshape, 22, 1.0)
    got, want = exec_pair(backend, ref_lib, _reduce_cmd("REDUCE_NORM2_FORWARD", *axis), nnc.NO_HINT, 0, [a], [np.zeros(oshape, F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-6)
    b = want[0]
    got, want = exec_pair(backend, ref_lib, _reduce_cmd("REDUCE_NORM2_BACKWARD", *axis), nnc.NO_HINT, 0, [g, a, b], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-6)
    got, want = exec_pair(backend, ref_lib, _reduce_cmd("REDUCE_NORM2_BACKWARD", *axis), nnc.NO_HINT, 0, [None, a, b], [np.zeros_like(a)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["MIN", "MAX"])
@pytest.mark.parametrize("shape", [(7,), (4, 5, 6, 3)])
def test_elementwise_min_max(backend, ref_lib, name, shape):
    a, b, g = _x(shape, 23, 1.0), _x(shape, 24, 1.0), _x(shape, 25, 1.0)
    b.flat[::5] = a.flat[::5]   # ties: the gradient goes to both
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd(name + "_FORWARD"), nnc.NO_HINT, 0, [a, b], [np.zeros_like(a)])
    assert np.array_equal(got[0], want[0])
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd(name + "_BACKWARD"), nnc.NO_HINT, 0, [g, a, b], [np.zeros_like(a), np.zeros_like(a)])
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd(name + "_BACKWARD"), nnc.NO_HINT, 0, [None, a, b], [np.zeros_like(a), np.zeros_like(a)])
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("name", ["ARGMAX", "ARGMIN"])
@pytest.mark.parametrize("shape,axis", [((5, 1000), 1), ((4, 6, 7), 1), ((9, 3), 0)])
@pytest.mark.parametrize("odt", [np.int32, F])
def test_argmax_argmin(backend, ref_lib, name, shape, axis, odt):
    a = _x(shape, 26, 1.0)
    a[tuple(0 for _ in shape)] = a.max() + 1 if name == "ARGMAX" else a.min() - 1
    oshape = tuple(1 if i == axis else d for i, d in enumerate(shape))
    got, want = exec_pair(backend, ref_lib, _reduce_cmd(name + "_FORWARD", axis), nnc.NO_HINT, 0, [a], [np.zeros(oshape, odt)])
    assert np.array_equal(got[0], want[0])


# ---- index select, pad (ccv_amd/csrc/cmd_index_pad.cpp) ------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(20, 16), (50,), (7, 130)])
def test_index_select(backend, ref_lib, shape):
    rng = np.random.default_rng(31)
    a = _x(shape, 32, 1.0)
    n = 11
    idx = rng.integers(0, shape[0], n).astype(np.int32)
    idx[3] = idx[7]   # repeated index: the scatter-add must accumulate
    oshape = (n,) + tuple(shape[1:])
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd("INDEX_SELECT_FORWARD"), nnc.NO_HINT, 0, [a, idx], [np.zeros(oshape, F)])
    assert np.array_equal(got[0], want[0])
    fidx = (rng.random(n) * (shape[0] - 1)).astype(F)   # fractional indices interpolate between neighbouring rows
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd("INDEX_SELECT_FORWARD"), nnc.NO_HINT, 0, [a, fidx], [np.zeros(oshape, F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-6, atol=1e-7)
    g = _x(oshape, 33, 1.0)
    got, want = exec_pair(backend, ref_lib, nnc.generic_cmd("INDEX_SELECT_BACKWARD"), nnc.NO_HINT, 0, [g, None, idx], [np.full(shape, 5, F)])
    assert np.array_equal(got[0], want[0])


@pytest.mark.parametrize("pad_type", [0, 1])
@pytest.mark.parametrize("shape,begin,end", [((5,), (2,), (3,)), ((3, 4), (1, 0), (0, 2)), ((2, 3, 4), (0, 1, 2), (1, 0, 1)), ((2, 3, 4, 5), (1, 1, 0, 2), (0, 2, 1, 1))])
def test_pad(backend, ref_lib, pad_type, shape, begin, end):
    a = _x(shape, 34, 1.0)
    oshape = tuple(d + b + e for d, b, e in zip(shape, begin, end))
    got, want = exec_pair(backend, ref_lib, nnc.CMD_PAD("PAD_FORWARD", pad_type, begin, end), nnc.NO_HINT, 0, [a], [np.full(oshape, 9, F)])
    assert np.array_equal(got[0], want[0])
    if pad_type == 0:   # the reference's backward is the crop of the zero-pad (pad_cpu_ref.c:88-138)
        g = _x(oshape, 35, 1.0)
        got, want = exec_pair(backend, ref_lib, nnc.CMD_PAD("PAD_BACKWARD", pad_type, begin, end), nnc.NO_HINT, 0, [g], [np.zeros(shape, F)])
        assert np.array_equal(got[0], want[0])


# ---- layer norm / rms norm over the trailing axes (ccv_amd/csrc/cmd_rownorm.cpp) -------------------------------------------------------
NORM_SHAPES = [((6, 40), (1,)), ((2, 5, 300), (2,)), ((3, 4, 5, 6), (2, 3)), ((4, 1030), (1,))]


@pytest.mark.parametrize("shape,axis", NORM_SHAPES)
@pytest.mark.parametrize("affine", [1, 0])
def test_layer_norm(backend, ref_lib, shape, axis, affine):
    a, g = _x(shape, 41, 2.0), _x(shape, 42, 1.0)
    sshape = tuple(1 if i in axis else d for i, d in enumerate(shape))       # statistics: one per row
    pshape = tuple(d if i in axis else 1 for i, d in enumerate(shape))       # scale / bias: one per normalised element
    scale, bias = _x(pshape, 43, 1.0) + 1.5, _x(pshape, 44, 1.0)
    ins = [a, scale, bias] if affine else [a]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_NORM("LAYER_NORM_FORWARD", 1e-5, affine, *axis), nnc.NO_HINT, 0, ins, [np.zeros_like(a), np.zeros(sshape, F), np.zeros(sshape, F)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=2e-5, atol=2e-6)
    mean, istd = want[1], want[2]
    if affine:
        bins = [g, None, None, a, scale, None, None, mean, istd]
        outs = [np.zeros_like(a), np.zeros(pshape, F), np.zeros(pshape, F)]
    else:
        bins = [g, None, None, a, None, mean, istd]
        outs = [np.zeros_like(a)]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_NORM("LAYER_NORM_BACKWARD", 1e-5, affine, *axis), nnc.NO_HINT, 0, bins, outs)
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape,axis", NORM_SHAPES)
def test_rmsnorm(backend, ref_lib, shape, axis):
    a, g = _x(shape, 45, 2.0), _x(shape, 46, 1.0)
    sshape = tuple(1 if i in axis else d for i, d in enumerate(shape))
    pshape = tuple(d if i in axis else 1 for i, d in enumerate(shape))
    scale = _x(pshape, 47, 1.0) + 1.5
    got, want = exec_pair(backend, ref_lib, nnc.CMD_NORM("RMSNORM_FORWARD", 1e-5, 0, *axis), nnc.NO_HINT, 0, [a, scale], [np.zeros_like(a), np.zeros(sshape, F)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=2e-5, atol=2e-6)
    istd = want[1]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_NORM("RMSNORM_BACKWARD", 1e-5, 0, *axis), nnc.NO_HINT, 0, [g, None, a, scale, None, istd], [np.zeros_like(a), np.zeros(pshape, F)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-5)


# ---- random fills (ccv_amd/csrc/cmd_ew.cpp): statistical parity, as the reference's own tests (test/int/nnc/random.tests.c) ---------------
def test_random_uniform_and_normal(backend):
    from harness import exec_on
    n = 200000
    r, out = exec_on(backend, nnc.GPU_MEMORY, nnc._blas_a("RANDOM_UNIFORM_FORWARD", -8.0, 4.0), nnc.NO_HINT, 0, [], [np.zeros(n, F)])
    assert r == 0
    u = out[0]
    # (the reference draws r in (0, 1) and rounds r u + (1 - r) l in fp32 as well: the ends can be met, not passed)
    assert u.min() >= -8 and u.max() <= 4 and abs(u.mean() + 2.0) < 0.05 and abs(u.std() - 12 / np.sqrt(12)) < 0.05
    assert len(np.unique(u)) > n * 0.95
    r, out2 = exec_on(backend, nnc.GPU_MEMORY, nnc._blas_a("RANDOM_UNIFORM_FORWARD", -8.0, 4.0), nnc.NO_HINT, 0, [], [np.zeros(n, F)])
    assert not np.array_equal(out2[0], u)   # a new seed per call
    r, out = exec_on(backend, nnc.GPU_MEMORY, nnc._blas_a("RANDOM_NORMAL_FORWARD", 2.0, 1.0), nnc.NO_HINT, 0, [], [np.zeros(n + 1, F)])  # odd count: last pair half used
    z = out[0]
    assert r == 0 and abs(z.mean() - 1.0) < 0.03 and abs(z.std() - 2.0) < 0.03
    assert abs(np.mean(((z - 1.0) / 2.0) ** 3)) < 0.05 and abs(np.mean(((z - 1.0) / 2.0) ** 4) - 3.0) < 0.15   # skewness, kurtosis


# ---- group norm (ccv_amd/csrc/cmd_groupnorm.cpp) ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt,shape,group_axis,groups,reduce_axis", [
    ("NCHW", (2, 16, 5, 7), 1, 4, (2, 3)),    # the UNet layout: channels in 4 groups, statistics over (group, H, W)
    ("NHWC", (2, 5, 7, 16), 3, 4, (1, 2)),
    ("NCHW", (3, 12, 4, 4), 1, 3, (2,)),      # reduce H only: one statistic per (n, group, x)
])
@pytest.mark.parametrize("affine", [1, 0])
def test_group_norm(backend, ref_lib, fmt, shape, group_axis, groups, reduce_axis, affine):
    a, g = _x(shape, 51, 2.0), _x(shape, 52, 1.0)
    sshape = tuple(groups if i == group_axis else (1 if i in reduce_axis else d) for i, d in enumerate(shape))
    pshape = tuple(d if i == group_axis else 1 for i, d in enumerate(shape))   # scale / bias per channel
    scale, bias = _x(pshape, 53, 1.0) + 1.5, _x(pshape, 54, 1.0)
    fwd = nnc.CMD_GROUP_NORM("GROUP_NORM_FORWARD", group_axis, groups, 1e-5, affine, *reduce_axis)
    bwd = nnc.CMD_GROUP_NORM("GROUP_NORM_BACKWARD", group_axis, groups, 1e-5, affine, *reduce_axis)
    ins = [a, scale, bias] if affine else [a]
    got, want = exec_pair(backend, ref_lib, fwd, nnc.NO_HINT, 0, ins, [np.zeros_like(a), np.zeros(sshape, F), np.zeros(sshape, F)], fmt=fmt)
    # (both reference backends read their epsilon through the lnorm member of the parameter union, i.e. ~0: kept, see cmd_groupnorm.cpp)
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=2e-5, atol=2e-6)
    mean, istd = want[1], want[2]
    if affine:
        bins, outs = [g, None, None, a, scale, None, None, mean, istd], [np.zeros_like(a), np.zeros(pshape, F), np.zeros(pshape, F)]
    else:
        bins, outs = [g, None, None, a, None, mean, istd], [np.zeros_like(a)]
    got, want = exec_pair(backend, ref_lib, bwd, nnc.NO_HINT, 0, bins, outs, fmt=fmt)
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(10,), (64, 3, 3, 3), (300, 77)])
def test_lamb(backend, ref_lib, shape):
    g, a, m = _x(shape, 61, 1.0), _x(shape, 62, 1.0), _x(shape, 63, 0.1)
    v = np.random.default_rng(64).random(shape, dtype=F) * 0.01
    cmd = nnc.CMD_ADAM_FORWARD(3, 0.002, 0.9, 0.98, 0.01, 1e-6, scale=0.5)   # lamb's parameters are adam's without amsgrad
    cmd.cmd = nnc.CMD["LAMB_FORWARD"]
    got, want = exec_pair(backend, ref_lib, cmd, nnc.NO_HINT, 0, [g, a, m, v], [np.zeros_like(a) for _ in range(3)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=3e-6, atol=1e-8)


# ---- upsample (ccv_amd/csrc/cmd_upsample.cpp): bit-exact, the taps and the summation order are the reference's ------------------------------
@pytest.mark.parametrize("fmt", ["NCHW", "NHWC"])
@pytest.mark.parametrize("up_type,align", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("src,dst", [((5, 7), (10, 14)), ((4, 6), (7, 9)), ((3, 3), (3, 3)), ((6, 5), (11, 10))])
def test_upsample(backend, ref_lib, fmt, up_type, align, src, dst):
    n, c = 2, 3
    ashape = (n, c) + src if fmt == "NCHW" else (n,) + src + (c,)
    bshape = (n, c) + dst if fmt == "NCHW" else (n,) + dst + (c,)
    a, g = _x(ashape, 71, 1.0), _x(bshape, 72, 1.0)
    ws, hs = dst[1] / src[1], dst[0] / src[0]
    got, want = exec_pair(backend, ref_lib, nnc.CMD_UPSAMPLE("UPSAMPLE_FORWARD", up_type, ws, hs, align), nnc.NO_HINT, 0, [a], [np.zeros(bshape, F)], fmt=fmt)
    assert np.array_equal(got[0], want[0])
    got, want = exec_pair(backend, ref_lib, nnc.CMD_UPSAMPLE("UPSAMPLE_BACKWARD", up_type, ws, hs, align), nnc.NO_HINT, 0, [g], [np.full(ashape, 5, F)], fmt=fmt)
    assert np.array_equal(got[0], want[0])


# The reference's four float bilinear int cases and the two half-precision ones (test/int/nnc/upsample.tests.c:15-172) read samples/chessbox.png and compare with a
# recorded file; the build has no libpng, so they are replayed here on a synthetic chessboard of the same kind -- hard edges, three channels, the reference's own scale
# factors (2 x up, 1/2 down) and both layouts -- against the reference's CPU backend instead of its recorded output (missing item 7).
def _chessboard(rows, cols):
    y, x = np.mgrid[0:rows, 0:cols]
    board = (((y // 12) + (x // 12)) % 2).astype(F)
    return np.stack([board * 255, 255 - board * 200, 40 + 100 * board + (x % 7)], axis=-1).astype(F)


@pytest.mark.parametrize("fmt", ["NHWC", "NCHW"])
@pytest.mark.parametrize("what", ["upsample", "downsample"])
@pytest.mark.parametrize("dtype", [np.float32, np.float16], ids=["float", "half"])
def test_bilinear_resample_of_a_chessboard_like_the_reference_int_cases(backend, ref_lib, fmt, what, dtype):
    """"upsample bilinear" = UPSAMPLE_FORWARD(BILINEAR, 2, 2, 0) of the image; "downsample bilinear" = UPSAMPLE_BACKWARD(BILINEAR, 2, 2, 0) with the image in the
    gradient's place (upsample.tests.c:63-172: the transposed operator is the reference's downsampler)."""
    rows, cols = 96, 128
    img = _chessboard(rows, cols)
    a = (img if fmt == "NHWC" else np.ascontiguousarray(img.transpose(2, 0, 1))).astype(dtype)
    orows, ocols = (rows * 2, cols * 2) if what == "upsample" else (rows // 2, cols // 2)
    bshape = (orows, ocols, 3) if fmt == "NHWC" else (3, orows, ocols)
    cmd = nnc.CMD_UPSAMPLE("UPSAMPLE_FORWARD" if what == "upsample" else "UPSAMPLE_BACKWARD", 1, 2, 2, 0)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [a], [np.zeros(bshape, dtype)], fmt)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [a.astype(F)], [np.zeros(bshape, F)], fmt, backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    if dtype == np.float32:
        assert np.array_equal(got[0], want[0])  # the taps and their order are the reference's
    else:
        np.testing.assert_allclose(got[0].astype(F), want[0], rtol=2e-3, atol=0.26)  # halves of values up to 4 x 255: one rounding of the fp32 result
