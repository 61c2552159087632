"""fp32 contractions on the bf16 matrix pipe with exactly split operands (ccv_amd/csrc/mfma_gemm_bf16x3.h, round 6).
The split x = hi + mid + lo is exact and all nine partial products are accumulated in fp32, so the results must meet the SAME bounds as the fp32 matrix
instructions' -- 1e-4 relative against the reference CPU backend (lib/nnc/cmd/blas/ccv_nnc_gemm_cpu_ref.c:110-184, lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:13-345),
and a few ulp against float64.  Both tile shapes, all four operand layouts, split-K, batched, ragged edges; the Winograd-via-HBM convolution (algorithm 1)
forward / data gradient / filter gradient whose 36 products are the kernel's customers in the VGG-D step."""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_on, exec_pair

F = np.float32


def srnd(rng, *shape, scale=1.0):
    return ((rng.random(shape, dtype=F) - 0.5) * 2 * scale).astype(F)


@pytest.fixture()
def split_mode(backend):
    prev = backend.tune_get("GEMM_BF16X3")
    yield backend
    backend.tune_set("GEMM_BF16X3", prev)


def test_split_is_exact_in_numpy():
    """The arithmetic the kernel relies on: clearing the low 16 bits twice leaves three bf16 values whose fp32 sum is x, bit for bit -- for |x| >= 2^-102
    (below that the last remainder's low bits fall under bf16's smallest denormal, 2^-133: an absolute error below 1e-40)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([srnd(rng, 100000, scale=s) for s in (1e-20, 1e-3, 1.0, 1e6, 1e30)] + [np.array([0.0, -0.0, 1.0, -1.0, np.float32(1) + np.float32(2) ** -23, 2.0 ** -102], F)])
    x = x[(x == 0) | (np.abs(x) >= 2.0 ** -102)]
    def top(v):
        return (v.view(np.uint32) & np.uint32(0xffff0000)).view(F)
    hi = top(x); r = x - hi; mid = top(r); lo = r - mid
    assert np.array_equal(top(lo), lo)                       # lo is a bf16 value already
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(F), x)
    tiny = srnd(rng, 100000, scale=1e-36)                    # ... and below: off by less than bf16's smallest denormal
    hi = top(tiny); r = tiny - hi; mid = top(r); lo = top(r - mid)
    assert np.abs(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64) - tiny.astype(np.float64)).max() < 2.0 ** -133


# (M, N, K, transpose_a, transpose_b): kk = both k-contiguous (NO_TRANSPOSE, TRANSPOSE), rr = both row-contiguous, and the two mixed forms
CASES = [
    (300, 260, 256, (0, 0), (0, 1)),   # kk, ragged edges of both tile shapes
    (512, 512, 96, (0, 0), (0, 1)),    # kk, K = three K-steps of 32 (six of the kernel's 16)
    (264, 384, 160, (0, 1), (0, 0)),   # rr (a stored [K][M], w stored [K][N])
    (256, 260, 128, (0, 0), (0, 0)),   # a k-contiguous, w row-contiguous
    (260, 256, 128, (0, 1), (0, 1)),   # a row-contiguous, w k-contiguous
]


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("case", CASES)
def test_gemm_forward_split(split_mode, ref_lib, case, mode):
    L = split_mode
    m, n, k, ta, tb = case
    rng = np.random.default_rng(1)
    a = srnd(rng, *((k, m) if ta != (0, 0) else (m, k)))
    w = srnd(rng, *((n, k) if tb != (0, 0) else (k, n)), scale=0.1)
    bias = srnd(rng, n)
    L.tune_set("GEMM_BF16X3", mode)
    got, want = exec_pair(L, ref_lib, nnc.CMD_GEMM_FORWARD(ta, tb), nnc.NO_HINT, 0, [a, w, bias], [np.zeros((m, n), F)])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-4)
    a64 = (a.T if ta != (0, 0) else a).astype(np.float64)
    w64 = (w.T if tb != (0, 0) else w).astype(np.float64)
    exact = a64 @ w64 + bias
    assert np.abs(got[0] - exact).max() <= 2e-6 * np.abs(exact).max()   # fp32-class accuracy: nothing of the 24 bits was dropped
    L.tune_set("GEMM_BF16X3", 0)
    (plain,) = exec_on(L, nnc.GPU_MEMORY, nnc.CMD_GEMM_FORWARD(ta, tb), nnc.NO_HINT, 0, [a, w, bias], [np.zeros((m, n), F)])[1]
    assert np.abs(plain - exact).max() <= 2e-6 * np.abs(exact).max()


@pytest.mark.parametrize("mode", [3, 4])
def test_gemm_backward_and_batched_split(split_mode, ref_lib, mode):
    L = split_mode
    rng = np.random.default_rng(2)
    m, n, k = 256, 320, 512
    a, w, g = srnd(rng, m, k), srnd(rng, n, k, scale=0.1), srnd(rng, m, n)
    L.tune_set("GEMM_BF16X3", mode)
    got, want = exec_pair(L, ref_lib, nnc.CMD_GEMM_BACKWARD((0, 0), (0, 1)), nnc.NO_HINT, 0, [g, a, w], [np.zeros_like(a), np.zeros_like(w), np.zeros(n, F)])
    for x, y in zip(got, want):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-4)
    # batched (grid z), accumulate flag
    a3, w3, old = srnd(rng, 3, 260, 128), srnd(rng, 3, 256, 128, scale=0.1), srnd(rng, 3, 260, 256)
    got, want = exec_pair(L, ref_lib, nnc.CMD_GEMM_FORWARD((0, 0), (1, 2)), nnc.NO_HINT, 0, [a3, w3], [old.copy()])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", [3, 4])
def test_winograd_via_hbm_on_the_split_contractions(split_mode, ref_lib, mode):
    """CONVOLUTION_FORWARD / BACKWARD with algorithm 1 (Winograd, V and M through HBM): 36 batched products (forward, data gradient: k-contiguous operands;
    filter gradient: row-contiguous operands, split-K) on the split kernel against the reference's direct loops, elementwise 1e-4 of the tensor's range."""
    L = split_mode
    rng = np.random.default_rng(3)
    n, h, w_, c, k = 2, 16, 16, 256, 256
    a, wt, b = srnd(rng, n, h, w_, c), srnd(rng, k, 3, 3, c, scale=1.0 / (9 * c) ** 0.5), srnd(rng, k, scale=0.05)
    g = srnd(rng, n, h, w_, k)
    hint = nnc.HINT((1, 1), (1, 1))
    L.tune_set("GEMM_BF16X3", mode)
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c)
    _, (want,) = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, h, w_, k), F)], backend=nnc.BACKEND_CPU_REF)
    cmd.algorithm = 1
    _, (got,) = exec_on(L, nnc.GPU_MEMORY, cmd, hint, 0, [a, wt, b], [np.zeros((n, h, w_, k), F)])
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    cmdb = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
    _, wants = exec_on(ref_lib, nnc.CPU_MEMORY, cmdb, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)], backend=nnc.BACKEND_CPU_REF)
    cmdb.algorithm = 1
    _, gots = exec_on(L, nnc.GPU_MEMORY, cmdb, hint, 0, [g, a, wt], [np.zeros_like(a), np.zeros_like(wt), np.zeros(k, F)])
    for x, y in zip(gots, wants):
        assert np.abs(x - y).max() <= 1e-4 * np.abs(y).max()
