"""BASELINE.json config 4 (ResNet-50, the reference's bin/nnc/imagenet.c trainer) at parity-test size: one bottleneck block
with a projection shortcut -- conv1x1 / BN / ReLU / conv3x3 (stride 1 or 2) / BN / ReLU / conv1x1 / BN, shortcut
(average-pool when strided) / conv1x1 / BN, EWSUM, ReLU -- forward and backward through the command interface, in **NCHW**
as that trainer runs it (bin/nnc/imagenet.c:354), against the reference's CPU backend.  The reference's CPU convolution
backward only exists for NHWC (lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:356-363), so the oracle chain runs the same
block in NHWC and the results are compared after transposition.  Covers the NCHW staging of conv (incl. Winograd behind
it), 1x1 and strided convs, BN training forward / backward in both layouts, EWSUM, average pooling and ReLU backward."""
import numpy as np
import pytest
from ccv_amd import nnc

F = np.float32


class Chain:
    def __init__(self, lib, mem, fmt, backend=None, pool_per_image=False):
        self.lib, self.mem, self.fmt, self.backend = lib, mem, fmt, backend
        # the reference's CPU pools only walk image 0 of a batch: the oracle chain issues them per image
        self.pool_per_image = pool_per_image

    def _pool(self, cmd, hint, ins_shapes, outs_shapes):
        """ins_shapes / outs_shapes: [(tensor, nhwc shape)]; one command, or one per image over 3-d aliases"""
        if not self.pool_per_image:
            return self.run(cmd, hint, [t for t, _ in ins_shapes], [t for t, _ in outs_shapes])
        n = ins_shapes[0][1][0]
        for i in range(n):
            al = lambda t, s: t.alias(s[1:], offset=i * s[1] * s[2] * s[3])
            self.run(cmd, hint, [al(t, s) for t, s in ins_shapes], [al(t, s) for t, s in outs_shapes])

    def t(self, arr_nhwc, kind="act"):
        """arr is given in NHWC logical order (weights: [K][kh][kw][C]); laid out in this chain's format"""
        a = np.asarray(arr_nhwc, F)
        if self.fmt == "NCHW" and a.ndim == 4:
            a = np.ascontiguousarray(a.transpose(0, 3, 1, 2))
        p = nnc.tensor_param(self.mem, nnc.NCHW if self.fmt == "NCHW" else nnc.NHWC, nnc.CCV_32F, a.shape)
        return self.lib.tensor(p, a)

    def np(self, t):
        a = t.numpy()
        if self.fmt == "NCHW" and a.ndim == 4:
            a = a.transpose(0, 2, 3, 1)
        return np.ascontiguousarray(a)

    def run(self, cmd, hint, ins, outs, flags=0):
        c = nnc.Cmd()
        nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
        if self.backend is not None:
            c.backend = self.backend
        r = self.lib.cmd_exec(c, hint, flags, ins, outs)
        assert r == 0, "command %x returned %d" % (cmd.cmd, r)

    def zeros_like_shape(self, nhwc_shape):
        return self.t(np.zeros(nhwc_shape, F))

    # --- layers (each returns output tensor + a closure for backward) ---
    def conv(self, x, xs, w_np, stride, pad):
        k, kh, kw, c = w_np.shape
        n, h, wd, _ = xs
        oh, ow = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
        w = self.t(w_np)
        y = self.zeros_like_shape((n, oh, ow, k))
        hint = nnc.HINT((stride, stride), (pad, pad))
        self.run(nnc.CMD_CONVOLUTION_FORWARD(1, k, kh, kw, c), hint, [x, w], [y])

        def back(g):
            dx, dw = self.zeros_like_shape(xs), self.t(np.zeros_like(w_np))
            self.run(nnc.CMD_CONVOLUTION_BACKWARD(1, k, kh, kw, c), hint, [g, x, w], [dx, dw])
            return dx, [self.np(dw)]
        return y, (n, oh, ow, k), back

    def bn(self, x, xs, scale_np, bias_np):
        c = xs[3]
        sshape = (1, 1, 1, c)
        axes = (0, 1, 2) if self.fmt == "NHWC" else (0, 2, 3)
        scale, bias = self.t(scale_np.reshape(sshape)), self.t(bias_np.reshape(sshape))
        mean, var = self.t(np.zeros(sshape, F)), self.t(np.ones(sshape, F))
        y, smean, sistd = self.zeros_like_shape(xs), self.t(np.zeros(sshape, F)), self.t(np.zeros(sshape, F))
        self.run(nnc.CMD_BATCH_NORM_FORWARD(1e-5, 0, 0.9, *axes), nnc.HINT(), [x, scale, bias, mean, var], [y, mean, var, smean, sistd])

        def back(g):
            dx, ds, db = self.zeros_like_shape(xs), self.t(np.zeros(sshape, F)), self.t(np.zeros(sshape, F))
            ins = [g, None, None, None, None, x, scale, None, None, None, None, None, None, smean, sistd]
            self.run(nnc.CMD_BATCH_NORM_BACKWARD(1e-5, 0, 0.9, *axes), nnc.HINT(), ins, [dx, ds, db])
            return dx, [self.np(ds).reshape(-1), self.np(db).reshape(-1)]
        return y, xs, back

    def relu(self, x, xs):
        y = self.zeros_like_shape(xs)
        self.run(nnc.CMD_RELU_FORWARD(), nnc.HINT(), [x], [y])

        def back(g):
            dx = self.zeros_like_shape(xs)
            self.run(nnc.CMD_RELU_BACKWARD(), nnc.HINT(), [g, None, y], [dx])
            return dx, []
        return y, xs, back

    def avgpool2(self, x, xs):
        n, h, w, c = xs
        y = self.zeros_like_shape((n, h // 2, w // 2, c))
        hint = nnc.HINT((2, 2), (0, 0))
        ys = (n, h // 2, w // 2, c)
        self._pool(nnc.CMD_AVERAGE_POOL_FORWARD(2, 2), hint, [(x, xs)], [(y, ys)])

        def back(g):
            dx = self.zeros_like_shape(xs)
            self._pool(nnc.CMD_AVERAGE_POOL_BACKWARD(2, 2), hint, [(g, ys)], [(dx, xs)])
            return dx, []
        return y, (n, h // 2, w // 2, c), back

    def add(self, a, b, s):
        y = self.zeros_like_shape(s)
        self.run(nnc.CMD_EWSUM_FORWARD(), nnc.HINT(), [a, b], [y])
        return y


def bottleneck(ch, x_np, g_np, P, stride, per_image_pool=False):
    """returns (y, dx, [param grads...]) as NHWC numpy arrays"""
    xs = x_np.shape
    x = ch.t(x_np)
    tape = []

    def push(res):
        y, s, back = res
        tape.append(back)
        return y, s
    # main branch
    y, s = push(ch.conv(x, xs, P["w1"], 1, 0))
    y, s = push(ch.bn(y, s, P["s1"], P["b1"]))
    y, s = push(ch.relu(y, s))
    y, s = push(ch.conv(y, s, P["w2"], stride, 1))
    y, s = push(ch.bn(y, s, P["s2"], P["b2"]))
    y, s = push(ch.relu(y, s))
    y, s = push(ch.conv(y, s, P["w3"], 1, 0))
    y3, s3 = push(ch.bn(y, s, P["s3"], P["b3"]))
    main_n = len(tape)
    # shortcut
    sc, ss = x, xs
    if stride == 2:
        sc, ss = push(ch.avgpool2(sc, ss))
    sc, ss = push(ch.conv(sc, ss, P["ws"], 1, 0))
    sc, ss = push(ch.bn(sc, ss, P["ss"], P["bs"]))
    assert ss == s3
    z = ch.add(y3, sc, s3)
    out, _, relu_back = ch.relu(z, s3)
    # backward
    g, _ = relu_back(ch.t(g_np))
    grads = []
    gm = g
    for back in reversed(tape[:main_n]):
        gm, pg = back(gm)
        grads = pg + grads
    gs = g
    sgrads = []
    for back in reversed(tape[main_n:]):
        gs, pg = back(gs)
        sgrads = pg + sgrads
    dx = ch.add(gm, gs, xs)
    return ch.np(out), ch.np(dx), grads + sgrads


@pytest.mark.parametrize("stride", [1, 2])
def test_resnet_bottleneck_nchw_matches_reference(backend, ref_lib, stride):
    rng = np.random.default_rng(3)
    n, h, c, m = 4, 8, 16, 8   # 16 -> (8, 8, 32) bottleneck on 8 x 8 maps
    u = lambda *s, sc=1.0: ((rng.random(s, dtype=F) - 0.5) * 2 * sc).astype(F)
    P = dict(w1=u(m, 1, 1, c, sc=0.3), w2=u(m, 3, 3, m, sc=0.2), w3=u(4 * m, 1, 1, m, sc=0.3), ws=u(4 * m, 1, 1, c, sc=0.3),
             s1=u(m) + 1.5, b1=u(m), s2=u(m) + 1.5, b2=u(m), s3=u(4 * m) + 1.5, b3=u(4 * m), ss=u(4 * m) + 1.5, bs=u(4 * m))
    x = u(n, h, h, c)
    oh = h // stride
    g = u(n, oh, oh, 4 * m)
    got = bottleneck(Chain(backend, nnc.GPU_MEMORY, "NCHW"), x, g, P, stride)
    want = bottleneck(Chain(ref_lib, nnc.CPU_MEMORY, "NHWC", backend=nnc.BACKEND_CPU_REF, pool_per_image=True), x, g, P, stride)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-5)
    scale = float(np.abs(want[1]).max())
    np.testing.assert_allclose(got[1], want[1], rtol=1e-3, atol=2e-5 * max(1.0, scale))
    assert len(got[2]) == len(want[2]) == 12
    for a, b in zip(got[2], want[2]):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=2e-5 * max(1.0, float(np.abs(b).max())))
