"""The seeded command cases behind tests/golden/nnc_golden.npz.  Inputs are regenerated from the seed (only the
reference OUTPUTS are stored), so the same table drives: the golden generator (reference CPU backend, build container),
the oracle pin test (plain-C restatement vs golden) and the GPU parity tests (HIP backend vs golden)."""
import numpy as np
from ccv_amd import nnc
from harness import exec_on, out_hw

F = np.float32


def _rng(name):
    return np.random.default_rng(abs(hash(name)) % (2**31) if False else sum(ord(c) * (i + 1) for i, c in enumerate(name)))


def _u(rng, *shape, scale=1.0):
    return ((rng.random(shape, dtype=F) - 0.5) * 2 * scale).astype(F)


def _conv(name, n, h, w, c, k, kh, kw, stride, border, groups=1, dil=None, bias=True, fmt="NHWC", backward=False, flags=0):
    def build():
        rng = _rng(name)
        hint = nnc.HINT(stride, border)
        ekh, ekw = ((kh - 1) * dil[0] + 1, (kw - 1) * dil[1] + 1) if dil else (kh, kw)
        oh, ow = out_hw(h, w, ekh, ekw, hint)
        a = _u(rng, n, h, w, c)
        wt = _u(rng, k, kh, kw, c // groups, scale=1.0 / (kh * kw * (c // groups)))
        b = _u(rng, k) if bias else None
        if fmt == "NCHW":
            a = np.ascontiguousarray(a.transpose(0, 3, 1, 2))
            wt = np.ascontiguousarray(wt.transpose(0, 3, 1, 2))
        if not backward:
            oshape = (n, oh, ow, k) if fmt == "NHWC" else (n, k, oh, ow)
            return dict(cmd=nnc.CMD_CONVOLUTION_FORWARD(groups, k, kh, kw, c // groups, dilation=dil), hint=hint, flags=flags, fmt=fmt,
                        inputs=[a, wt] + ([b] if bias else []), outputs=[np.zeros(oshape, F)])
        g = _u(rng, n, oh, ow, k)
        return dict(cmd=nnc.CMD_CONVOLUTION_BACKWARD(groups, k, kh, kw, c // groups, dilation=dil), hint=hint, flags=flags, fmt=fmt,
                    inputs=[g, a, wt], outputs=[np.zeros_like(a), _u(rng, *wt.shape), np.zeros(k, F)])
    return build


def _gemm(name, m, n, k, ta=False, tb=False, bias=True, batch=0, backward=False, flags=0):
    def build():
        rng = _rng(name)
        pre = (batch,) if batch else ()
        a = _u(rng, *(pre + ((k, m) if ta else (m, k))))
        w = _u(rng, *((n, k) if tb else (k, n)), scale=1.0 / k)
        b = _u(rng, n) if bias else None
        nd = len(pre) + 2
        tra = (nd - 2, nd - 1) if ta else (0, 0)
        trb = (0, 1) if tb else (0, 0)
        if not backward:
            return dict(cmd=nnc.CMD_GEMM_FORWARD(tra, trb), hint=nnc.HINT(), flags=flags, fmt="NHWC",
                        inputs=[a, w] + ([b] if bias else []), outputs=[np.zeros(pre + (m, n), F)])
        g = _u(rng, *(pre + (m, n)))
        return dict(cmd=nnc.CMD_GEMM_BACKWARD(tra, trb), hint=nnc.HINT(), flags=flags, fmt="NHWC",
                    inputs=[g, a, w], outputs=[np.zeros_like(a), _u(rng, *w.shape), _u(rng, n)])
    return build


def _pool(name, kind, n, h, w, c, k, stride, border, backward=False):
    def build():
        rng = _rng(name)
        hint = nnc.HINT(stride, border)
        oh, ow = out_hw(h, w, k[0], k[1], hint)
        a = _u(rng, n, h, w, c)
        if kind == "max":  # plant ties so the every-maximum-gets-the-gradient rule is exercised
            a = np.round(a * 4) / 4
        fwd = nnc.CMD_MAX_POOL_FORWARD(*k) if kind == "max" else nnc.CMD_AVERAGE_POOL_FORWARD(*k)
        if not backward:
            return dict(cmd=fwd, hint=hint, flags=0, fmt="NHWC", inputs=[a], outputs=[np.zeros((n, oh, ow, c), F)], pool=True)
        g = _u(rng, n, oh, ow, c)
        bwd = nnc.CMD_MAX_POOL_BACKWARD(*k) if kind == "max" else nnc.CMD_AVERAGE_POOL_BACKWARD(*k)
        return dict(cmd=bwd, hint=hint, flags=0, fmt="NHWC", inputs=[g, a, None], outputs=[np.zeros_like(a)], pool=True, pool_fwd=fwd, pool_oshape=(n, oh, ow, c))
    return build


def _smce(name, n, c, label="f32", smooth=None, backward=False):
    def build():
        rng = _rng(name)
        a = _u(rng, n, c, scale=3)
        lab = rng.integers(0, c, n)
        if label == "f32":
            b = lab.astype(F)
        elif label == "i32":
            b = lab.astype(np.int32)
        else:
            b = rng.random((n, c), dtype=F)
            b = (b / b.sum(1, keepdims=True)).astype(F)
        t0, t1 = smooth if smooth else (0.0, 1.0)
        if not backward:
            return dict(cmd=nnc.CMD_SOFTMAX_CROSSENTROPY_FORWARD(t0, t1), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[a, b], outputs=[np.zeros(n, F), np.zeros((n, c), F)])
        e = np.exp(a - a.max(1, keepdims=True))
        d = (e / e.sum(1, keepdims=True)).astype(F)
        g = _u(rng, n)
        return dict(cmd=nnc.CMD_SOFTMAX_CROSSENTROPY_BACKWARD(t0, t1), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[g, None, None, b, None, d], outputs=[np.zeros((n, c), F)])
    return build


def _simple(name, kind):
    def build():
        rng = _rng(name)
        x = _u(rng, 3, 5, 7, 6)
        if kind == "relu":
            return dict(cmd=nnc.CMD_RELU_FORWARD(), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[x], outputs=[np.zeros_like(x)])
        if kind == "relu_back":
            return dict(cmd=nnc.CMD_RELU_BACKWARD(), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[_u(rng, *x.shape), None, np.maximum(x, 0)], outputs=[np.zeros_like(x)])
        if kind == "ewsum":
            return dict(cmd=nnc.CMD_EWSUM_FORWARD(), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[x, _u(rng, *x.shape), _u(rng, *x.shape), _u(rng, *x.shape)], outputs=[np.zeros_like(x)])
        if kind == "scalar_mul":
            return dict(cmd=nnc.CMD_SCALAR_MUL_FORWARD(0.3), hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[x], outputs=[np.zeros_like(x)])
        if kind.startswith("sgd"):
            nest = int(kind.endswith("nesterov"))
            return dict(cmd=nnc.CMD_SGD_FORWARD(nest, 0.01, 0.5, 0.0005, 0.9, 0.0 if nest else 0.9), hint=nnc.HINT(), flags=0, fmt="NHWC",
                        inputs=[_u(rng, *x.shape), x, _u(rng, *x.shape)], outputs=[np.zeros_like(x), np.zeros_like(x)])
        raise KeyError(kind)
    return build


def _bnorm(name, shape, fmt, is_test=False, backward=False):
    def build():
        rng = _rng(name)
        x = _u(rng, *shape, scale=2.0)
        caxis = 3 if fmt == "NHWC" else 1
        C = shape[caxis]
        sshape = tuple(C if k == caxis else 1 for k in range(4))
        axes = tuple(k for k in range(4) if k != caxis)
        scale, bias = _u(rng, *sshape) + 1.5, _u(rng, *sshape)
        mean, var = _u(rng, *sshape), rng.random(sshape, dtype=F) + 0.5
        if not backward:
            outs = [np.zeros_like(x)] if is_test else [np.zeros_like(x), "in3", "in4", np.zeros(sshape, F), np.zeros(sshape, F)]
            return dict(cmd=nnc.CMD_BATCH_NORM_FORWARD(1e-4, int(is_test), 0.9, *axes), hint=nnc.HINT(), flags=0, fmt=fmt, inputs=[x, scale, bias, mean, var], outputs=outs)
        red = tuple(axes)
        mu = x.mean(axis=red, keepdims=True).astype(F)
        istd = (1.0 / np.sqrt(((x - mu) ** 2).mean(axis=red, keepdims=True) + 1e-4)).astype(F)
        g = _u(rng, *shape)
        ins = [g] + [None] * 4 + [x, scale] + [None] * 6 + [mu, istd]
        return dict(cmd=nnc.CMD_BATCH_NORM_BACKWARD(1e-4, 0, 0.9, *axes), hint=nnc.HINT(), flags=0, fmt=fmt, inputs=ins, outputs=[np.zeros_like(x), np.zeros(sshape, F), np.zeros(sshape, F)])
    return build


def _bcast(name, kind, ashape, bshape, backward=False, with_g=True):
    def build():
        rng = _rng(name)
        a, b = _u(rng, *ashape), _u(rng, *bshape)
        cshape = np.broadcast_shapes(ashape, bshape)
        if not backward:
            cmd = nnc.CMD_ADD_FORWARD(0.5, 0.3) if kind == "add" else nnc.CMD_MUL_FORWARD(0.7)
            return dict(cmd=cmd, hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[a, b], outputs=[np.zeros(cshape, F)])
        g = _u(rng, *cshape) if with_g else None
        cmd = nnc.CMD_ADD_BACKWARD(0.5, 0.3) if kind == "add" else nnc.CMD_MUL_BACKWARD(0.7)
        return dict(cmd=cmd, hint=nnc.HINT(), flags=0, fmt="NHWC", inputs=[g, a, b], outputs=[np.zeros(ashape, F), np.zeros(bshape, F)])
    return build


CASES = {}


def _add(name, builder):
    CASES[name] = builder


_add("conv_fwd_3x3_c8", _conv("conv_fwd_3x3_c8", 2, 9, 10, 8, 16, 3, 3, (1, 1), (1, 1)))
_add("conv_fwd_5x5_s2_c3", _conv("conv_fwd_5x5_s2_c3", 1, 12, 11, 3, 5, 5, 5, (2, 2), (2, 2)))
_add("conv_fwd_1x1_nobias", _conv("conv_fwd_1x1_nobias", 2, 7, 7, 12, 8, 1, 1, (1, 1), (0, 0), bias=False))
_add("conv_fwd_7x7_s2", _conv("conv_fwd_7x7_s2", 1, 15, 13, 4, 6, 7, 7, (2, 2), (3, 3)))
_add("conv_fwd_groups2", _conv("conv_fwd_groups2", 2, 8, 8, 8, 8, 3, 3, (1, 1), (1, 1), groups=2))
_add("conv_fwd_dilation2", _conv("conv_fwd_dilation2", 1, 11, 11, 4, 4, 3, 3, (1, 1), (2, 2), dil=(2, 2)))
_add("conv_fwd_multi_tile", _conv("conv_fwd_multi_tile", 3, 6, 20, 40, 130, 3, 3, (1, 1), (1, 1)))
_add("conv_fwd_nchw", _conv("conv_fwd_nchw", 2, 9, 10, 8, 16, 3, 3, (1, 1), (1, 1), fmt="NCHW"))
_add("conv_fwd_vgg_first", _conv("conv_fwd_vgg_first", 1, 33, 33, 3, 64, 3, 3, (1, 1), (0, 0)))
_add("conv_bwd_3x3_c8", _conv("conv_bwd_3x3_c8", 2, 9, 10, 8, 16, 3, 3, (1, 1), (1, 1), backward=True))
_add("conv_bwd_5x5_s2_c3", _conv("conv_bwd_5x5_s2_c3", 1, 12, 11, 3, 5, 5, 5, (2, 2), (2, 2), backward=True))
_add("conv_bwd_groups2", _conv("conv_bwd_groups2", 2, 8, 8, 8, 8, 3, 3, (1, 1), (1, 1), groups=2, backward=True))
_add("conv_bwd_dilation2", _conv("conv_bwd_dilation2", 1, 11, 11, 4, 4, 3, 3, (1, 1), (2, 2), dil=(2, 2), backward=True))
_add("conv_bwd_multi_tile_acc", _conv("conv_bwd_multi_tile_acc", 3, 6, 20, 40, 130, 3, 3, (1, 1), (1, 1), backward=True, flags=nnc.ACCUMULATE_OUTPUT))
_add("gemm_fwd", _gemm("gemm_fwd", 10, 20, 33))
_add("gemm_fwd_ta", _gemm("gemm_fwd_ta", 10, 20, 33, ta=True, bias=False))
_add("gemm_fwd_tb", _gemm("gemm_fwd_tb", 37, 130, 70, tb=True))
_add("gemm_fwd_ta_tb", _gemm("gemm_fwd_ta_tb", 12, 9, 16, ta=True, tb=True))
_add("gemm_fwd_batch2", _gemm("gemm_fwd_batch2", 6, 8, 10, batch=2))
_add("gemm_bwd", _gemm("gemm_bwd", 10, 20, 33, backward=True))
_add("gemm_bwd_tb", _gemm("gemm_bwd_tb", 37, 130, 70, tb=True, backward=True))
_add("gemm_bwd_tb_acc", _gemm("gemm_bwd_tb_acc", 16, 24, 40, tb=True, backward=True, flags=nnc.ACCUMULATE_OUTPUT))
_add("gemm_bwd_batch2", _gemm("gemm_bwd_batch2", 6, 8, 10, batch=2, backward=True))
_add("maxpool_fwd_3x3_s2", _pool("maxpool_fwd_3x3_s2", "max", 2, 11, 13, 5, (3, 3), (2, 2), (0, 0)))
_add("maxpool_fwd_3x3_s2_b1", _pool("maxpool_fwd_3x3_s2_b1", "max", 1, 12, 12, 4, (3, 3), (2, 2), (1, 1)))
_add("maxpool_fwd_2x2_s2", _pool("maxpool_fwd_2x2_s2", "max", 2, 8, 10, 3, (2, 2), (2, 2), (0, 0)))
_add("maxpool_bwd_3x3_s2", _pool("maxpool_bwd_3x3_s2", "max", 2, 11, 13, 5, (3, 3), (2, 2), (0, 0), backward=True))
_add("maxpool_bwd_3x3_s2_b1", _pool("maxpool_bwd_3x3_s2_b1", "max", 1, 12, 12, 4, (3, 3), (2, 2), (1, 1), backward=True))
_add("avgpool_fwd_3x3_s2_b1", _pool("avgpool_fwd_3x3_s2_b1", "avg", 2, 12, 12, 4, (3, 3), (2, 2), (1, 1)))
_add("avgpool_fwd_2x2_s2", _pool("avgpool_fwd_2x2_s2", "avg", 2, 8, 10, 3, (2, 2), (2, 2), (0, 0)))
_add("avgpool_bwd_3x3_s2_b1", _pool("avgpool_bwd_3x3_s2_b1", "avg", 2, 12, 12, 4, (3, 3), (2, 2), (1, 1), backward=True))
_add("smce_fwd_f32label", _smce("smce_fwd_f32label", 6, 50))
_add("smce_fwd_i32label", _smce("smce_fwd_i32label", 6, 50, label="i32"))
_add("smce_fwd_onehot", _smce("smce_fwd_onehot", 5, 20, label="dense"))
_add("smce_fwd_smooth", _smce("smce_fwd_smooth", 6, 50, smooth=(0.002, 0.9)))
_add("smce_bwd_f32label", _smce("smce_bwd_f32label", 6, 50, backward=True))
_add("smce_bwd_smooth", _smce("smce_bwd_smooth", 6, 50, smooth=(0.002, 0.9), backward=True))
_add("bnorm_fwd_nhwc", _bnorm("bnorm_fwd_nhwc", (3, 5, 4, 10), "NHWC"))
_add("bnorm_fwd_nchw", _bnorm("bnorm_fwd_nchw", (3, 6, 5, 4), "NCHW"))
_add("bnorm_fwd_test_nhwc", _bnorm("bnorm_fwd_test_nhwc", (2, 4, 4, 70), "NHWC", is_test=True))
_add("bnorm_bwd_nhwc", _bnorm("bnorm_bwd_nhwc", (3, 5, 4, 10), "NHWC", backward=True))
_add("bnorm_bwd_nchw", _bnorm("bnorm_bwd_nchw", (3, 6, 5, 4), "NCHW", backward=True))
_add("add_fwd_bcast", _bcast("add_fwd_bcast", "add", (2, 3, 4), (4,)))
_add("add_fwd_same", _bcast("add_fwd_same", "add", (2, 3, 4, 5), (2, 3, 4, 5)))
_add("mul_fwd_bcast", _bcast("mul_fwd_bcast", "mul", (4, 1), (2,)))
_add("add_bwd_bcast", _bcast("add_bwd_bcast", "add", (2, 3, 4), (4,), backward=True))
_add("mul_bwd_bcast", _bcast("mul_bwd_bcast", "mul", (4, 1), (1, 2), backward=True))
_add("mul_bwd_same", _bcast("mul_bwd_same", "mul", (2, 3, 4), (2, 3, 4), backward=True))
for _k in ("relu", "relu_back", "ewsum", "scalar_mul", "sgd", "sgd_nesterov"):
    _add(_k, _simple(_k, _k))


def build_case(name):
    return CASES[name]()


def _exec_inplace(lib, mem, case, backend):
    """Like harness.exec_on, but outputs given as "inN" alias input N's tensor (batch norm updates its running statistics in place)."""
    from harness import make_tensors
    c = nnc.Cmd()
    nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(case["cmd"]), nnc.C.sizeof(c))
    if backend is not None:
        c.backend = backend
    it = make_tensors(lib, mem, case["inputs"], case["fmt"])
    ot = [it[int(o[2:])] if isinstance(o, str) else make_tensors(lib, mem, [o], case["fmt"])[0] for o in case["outputs"]]
    ret = lib.cmd_exec(c, case["hint"], case["flags"], it, ot)
    assert ret == 0, "exec returned %d" % ret
    return [t.numpy() for t in ot]


def run_case(lib, mem, case, backend=None, per_image_pool=False):
    """Execute a case on `lib`; returns the list of output arrays.  Max-pool backward needs the forward output as input 2:
    it is produced by the same library first.  per_image_pool: issue pools image by image (reference CPU pools)."""
    def one(cmd, inputs, outputs, fmt):
        ret, res = exec_on(lib, mem, cmd, case["hint"], case["flags"], inputs, outputs, fmt, backend=backend)
        assert ret == 0, "exec returned %d" % ret
        return res
    if any(isinstance(o, str) for o in case["outputs"]):
        return _exec_inplace(lib, mem, case, backend)
    if not case.get("pool"):
        return one(case["cmd"], case["inputs"], case["outputs"], case["fmt"])
    n = case["inputs"][0].shape[0]
    images = range(n) if per_image_pool else [slice(None)]
    outs = [np.zeros_like(o) for o in case["outputs"]]
    for im in images:
        sel = (lambda x: None if x is None else (x[im] if per_image_pool else x))
        ins = [sel(x) for x in case["inputs"]]
        if "pool_fwd" in case:  # backward: inputs (g, a, b = forward(a))
            shp = case["pool_oshape"][1:] if per_image_pool else case["pool_oshape"]
            ins[2] = one(case["pool_fwd"], [ins[1]], [np.zeros(shp, F)], case["fmt"])[0]
        res = one(case["cmd"], ins, [sel(o).copy() for o in case["outputs"]], case["fmt"])
        for o, r in zip(outs, res):
            o[im] = r
    return outs
