"""Seeded random geometries for the convolution rows -- kernel size, stride, padding, dilation, groups, channel counts that are
and are not multiples of the vector / tile widths, tiny and ragged maps -- forward and backward, under the backend's own algorithm
choice and under each explicit one, against the reference's CPU backend.  (The fixed cases of test_parity_ops.py pin the paths;
this sweeps the space between them.)"""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_on, out_hw

F = np.float32


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 2, 3, 3, 3, 5]))
        stride = int(rng.choice([1, 1, 1, 2]))
        dil = int(rng.choice([1, 1, 1, 2])) if k > 1 else 1
        pad = int(rng.integers(0, (k - 1) * dil // 2 + 2))
        groups = int(rng.choice([1, 1, 1, 2]))
        c = int(rng.choice([1, 3, 4, 8, 12, 32, 36, 64])) * groups
        ko = int(rng.choice([1, 4, 8, 20, 32, 48, 64])) * groups
        h, w = int(rng.integers(1, 15)), int(rng.integers(1, 15))
        nb = int(rng.integers(1, 4))
        ek = (k - 1) * dil + 1
        if h + 2 * pad < ek or w + 2 * pad < ek or pad > ek - 1:
            continue
        out.append((nb, h, w, c, ko, k, stride, pad, dil, groups, bool(rng.integers(0, 2))))
    return out


CASES = _cases(16, 2024)


@pytest.mark.parametrize("case", CASES, ids=["n%d_%dx%d_c%d_k%d_%dx%d_s%d_p%d_d%d_g%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[5], c[6], c[7], c[8], c[9], "_b" if c[10] else "") for c in CASES])
def test_random_conv_geometry(backend, ref_lib, case):
    nb, h, w, c, ko, k, stride, pad, dil, groups, bias = case
    rng = np.random.default_rng(hash(case) % (1 << 31))
    sr = lambda *s, scale=1.0: ((rng.random(s, dtype=F) - 0.5) * 2 * scale).astype(F)
    a, wt = sr(nb, h, w, c), sr(ko, k, k, c // groups, scale=1.0 / max(1, k * k * (c // groups)) ** 0.5)
    b = sr(ko) if bias else None
    hint = nnc.HINT((stride, stride), (pad, pad))
    ek = (k - 1) * dil + 1
    oh, ow = out_hw(h, w, ek, ek, hint)
    g = sr(nb, oh, ow, ko)
    d = (dil, dil) if dil > 1 else None
    fwd, bwd = nnc.CMD_CONVOLUTION_FORWARD(groups, ko, k, k, c // groups, dilation=d), nnc.CMD_CONVOLUTION_BACKWARD(groups, ko, k, k, c // groups, dilation=d)
    ins = [a, wt] + ([b] if bias else [])
    r, want_f = exec_on(ref_lib, nnc.CPU_MEMORY, fwd, hint, 0, ins, [np.zeros((nb, oh, ow, ko), F)], backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    r, want_b = exec_on(ref_lib, nnc.CPU_MEMORY, bwd, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(ko, F)], backend=nnc.BACKEND_CPU_REF)
    assert r == 0
    tol = lambda ref: dict(rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    for algo in (-1, 0, 1):
        fwd.algorithm = bwd.algorithm = algo
        r, got = exec_on(backend, nnc.GPU_MEMORY, fwd, hint, 0, ins, [np.full((nb, oh, ow, ko), 3, F)])
        assert r == 0
        np.testing.assert_allclose(got[0], want_f[0], **tol(want_f[0]))
        r, got = exec_on(backend, nnc.GPU_MEMORY, bwd, hint, 0, [g, a, wt], [np.full_like(a, 3), np.zeros_like(wt), np.zeros(ko, F)])
        assert r == 0
        for x, y in zip(got, want_b):
            np.testing.assert_allclose(x, y, **tol(y))
    fwd.algorithm = bwd.algorithm = -1
