import numpy as np
from ccv_amd import nnc


def _conv_ref(ref, a, w, bias, hint, kh, kw, K, groups=1):
    at = ref.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, *a.shape), a)
    wt = ref.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, *w.shape), w)
    bt = ref.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, K), bias)
    oh = (a.shape[1] + hint.border.begin[0] + hint.border.end[0] - kh) // hint.stride.dim[0] + 1
    ow = (a.shape[2] + hint.border.begin[1] + hint.border.end[1] - kw) // hint.stride.dim[1] + 1
    ot = ref.tensor(nnc.CPU_TENSOR_NHWC(nnc.CCV_32F, a.shape[0], oh, ow, K))
    cmd = nnc.CMD_CONVOLUTION_FORWARD(groups, K, kh, kw, a.shape[3] // groups)
    cmd.backend = nnc.BACKEND_CPU_REF
    assert ref.cmd_exec(cmd, hint, 0, [at, wt, bt], [ot]) == 0
    return ot.numpy()


def test_conv_fwd_small(backend, ref_lib):
    rng = np.random.default_rng(0)
    a = rng.random((2, 9, 10, 8), dtype=np.float32)
    w = rng.random((16, 3, 3, 8), dtype=np.float32) / 72
    bias = rng.random(16, dtype=np.float32)
    hint = nnc.HINT((1, 1), (1, 1))
    want = _conv_ref(ref_lib, a, w, bias, hint, 3, 3, 16)
    L = backend
    at = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, *a.shape), a)
    wt = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, *w.shape), w)
    bt = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, 16), bias)
    ot = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, *want.shape))
    assert L.cmd_exec(nnc.CMD_CONVOLUTION_FORWARD(1, 16, 3, 3, 8), hint, 0, [at, wt, bt], [ot]) == 0
    got = ot.numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_graft_entry_smoke_body(emu_lib):
    """__graft_entry__.smoke() itself (the driver runs it on the MI355X at round end), here on the emulator build."""
    import __graft_entry__ as g
    g.smoke(emu_lib)
