"""Shared helpers: run one nnc command on the backend under test (GPU tensors) and on the reference's own CPU
backend (oracle/_ref, CPU tensors) from the same numpy inputs."""
import numpy as np
from ccv_amd import nnc

_FMT = {"NHWC": nnc.NHWC, "NCHW": nnc.NCHW}


def _param(mem, fmt, arr, device=0):
    dt = nnc._NP_DT[np.dtype(arr.dtype)]
    return nnc.tensor_param(mem, _FMT[fmt] if isinstance(fmt, str) else fmt, dt, arr.shape, device if mem == nnc.GPU_MEMORY else 0)


def make_tensors(lib, mem, arrays, fmt="NHWC"):
    out = []
    for a in arrays:
        if a is None:
            out.append(None)
        else:
            a = np.asarray(a)
            out.append(lib.tensor(_param(mem, fmt, a), a))
    return out


def exec_on(lib, mem, cmd, hint, flags, inputs, outputs, fmt="NHWC", backend=None):
    """inputs / outputs: lists of numpy arrays (outputs give shape + initial contents) or None. Returns output arrays."""
    c = nnc.Cmd()
    nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
    if backend is not None:
        c.backend = backend
    it = make_tensors(lib, mem, inputs, fmt)
    ot = make_tensors(lib, mem, outputs, fmt)
    ret = lib.cmd_exec(c, hint, flags, it, ot)
    res = [t.numpy() if t is not None else None for t in ot]
    return ret, res


def exec_pair(L, ref, cmd, hint, flags, inputs, outputs, fmt="NHWC", ref_backend=nnc.BACKEND_CPU_REF):
    r1, got = exec_on(L, nnc.GPU_MEMORY, cmd, hint, flags, inputs, outputs, fmt)
    r2, want = exec_on(ref, nnc.CPU_MEMORY, cmd, hint, flags, inputs, outputs, fmt, backend=ref_backend)
    assert r2 == 0, "reference returned %d" % r2
    assert r1 == 0, "backend returned %d" % r1
    return got, want


def out_hw(h, w, kh, kw, hint):
    oh = (h + hint.border.begin[0] + hint.border.end[0] - kh) // max(1, hint.stride.dim[0]) + 1
    ow = (w + hint.border.begin[1] + hint.border.end[1] - kw) // max(1, hint.stride.dim[1]) + 1
    return oh, ow


def tensor_eq(a, b):
    """REQUIRE_TENSOR_EQ semantics (lib/nnc/ccv_nnc_tensor.c:473-536): fail only if |ulp diff| > 128 AND |a-b| > FLT_EPSILON."""
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    b = np.ascontiguousarray(b, dtype=np.float32).ravel()
    if a.shape != b.shape:
        return False
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(2**31) - ai, ai)
    bi = np.where(bi < 0, -(2**31) - bi, bi)
    bad = (np.abs(ai - bi) > 128) & (np.abs(a - b) > np.finfo(np.float32).eps)
    return not bad.any()
