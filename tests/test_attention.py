"""SCALED_DOT_PRODUCT_ATTENTION forward / backward (cmd_attention.cpp) against the reference's CPU implementation
(lib/nnc/cmd/scaled_dot_product_attention/ccv_nnc_scaled_dot_product_attention_cpu_ref.c) in fp32 on the same inputs:
grouped-query heads, causal rows that see no key (R > C), an additive mask broadcast over batch / heads, one-head 3-d tensors,
head sizes in each of the kernel's three LDS layouts, and the head-unifying projection.  Tolerance: 1e-4 relative to the
tensor's largest value (north_star's fp32 bound); the reference's own GPU test uses 3e-3 (half precision, cublas.tests.c:2816)."""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import exec_on

F = np.float32


def sdpa_cmd(name, scale, causal):
    c = nnc.generic_cmd(name)
    c.info.f1.v = scale          # scaled_dot_product_attention.scale is the first field of the parameter union
    c.info.blas.transpose_a[1] = int(causal)  # .is_causal, the second int-sized field
    return c


def close(got, want, tol=1e-4):
    bound = tol * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= bound, "max |diff| %.3g > %.3g" % (err, bound)


CASES = [
    # B, R, C, Hq, Hk, D, Dv, causal, mask
    (2, 20, 37, 4, 4, 24, 24, False, None),
    (2, 37, 20, 4, 2, 16, 40, True, None),       # R > C: the first rows see no key at all; grouped-query heads; Dv != D
    (1, 33, 33, 3, 3, 72, 72, True, None),       # 64 < D <= 128
    (1, 18, 50, 2, 1, 136, 136, False, None),    # D > 128
    (2, 19, 21, 4, 4, 32, 32, False, (1, 1)),    # one mask for every batch item and head
    (2, 19, 21, 4, 2, 32, 32, True, (2, 4)),     # a mask per batch item and head, and causal
    # the matrix-core forward kernel (D % 8 == 0, Dv % 32 == 0, both <= 128): several waves and workgroups, ragged row / key blocks
    (2, 150, 70, 4, 2, 64, 64, True, None),
    (1, 131, 97, 2, 2, 128, 96, False, (1, 2)),
    (2, 40, 200, 3, 1, 40, 32, True, (2, 1)),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_attention_forward_backward(backend, ref_lib, case):
    B, R, Cn, Hq, Hk, D, Dv, causal, mask_shape = case
    rng = np.random.default_rng(5)
    q = (rng.random((B, R, Hq, D), dtype=F) - F(0.5))
    k = (rng.random((B, Cn, Hk, D), dtype=F) - F(0.5))
    v = (rng.random((B, Cn, Hk, Dv), dtype=F) - F(0.5))
    mask = None if mask_shape is None else (rng.random(mask_shape + (R, Cn), dtype=F) - F(0.5)) * F(2)
    scale = 1.0 / np.sqrt(D)
    fcmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD", scale, causal)
    ins = [q, k, v] + ([mask] if mask is not None else [])
    r1, got = exec_on(backend, nnc.GPU_MEMORY, fcmd, nnc.NO_HINT, 0, ins, [np.zeros((B, R, Hq, Dv), F), np.zeros((B, Hq, R), F)])
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, nnc.NO_HINT, 0, ins, [np.zeros((B, R, Hq, Dv), F), np.zeros((B, Hq, R), F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    close(got[0], want[0])
    # log-sum-exp rows (the CPU reference does not write them): check against numpy for the rows that see a key
    s = np.einsum("brhd,bchd->bhrc", q, np.repeat(k, Hq // Hk, axis=2)).astype(np.float64) * scale
    if mask is not None:
        s = s + mask
    if causal:
        vis = np.arange(R)[:, None] - R + Cn + 1
        s = np.where(np.arange(Cn)[None, :] < vis, s, -np.inf)
    with np.errstate(divide="ignore"):
        lse = np.log(np.exp(s - s.max(-1, keepdims=True).clip(-1e30)).sum(-1)) + s.max(-1).clip(-1e30)
    seen = np.isfinite(s).any(-1)
    np.testing.assert_allclose(got[1][seen], lse[seen], rtol=1e-4, atol=1e-5)
    g = (rng.random((B, R, Hq, Dv), dtype=F) - F(0.5))
    bcmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_BACKWARD", scale, causal)
    bins = [g, None, None, q, k, v] + ([mask] if mask is not None else [])
    outs = [np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)]
    r1, gb = exec_on(backend, nnc.GPU_MEMORY, bcmd, nnc.NO_HINT, 0, bins, outs)
    assert r1 == 0
    # the gradients in closed form (float64): p = softmax(s), dS = p * (dP - sum_y p dP)
    ratio = Hq // Hk
    kr, vr = np.repeat(k, ratio, axis=2).astype(np.float64), np.repeat(v, ratio, axis=2).astype(np.float64)
    with np.errstate(invalid="ignore"):
        p = np.exp(s - np.where(seen, s.max(-1), 0.0)[..., None])
    p = np.where(np.isfinite(s), p, 0.0)
    p = p / np.where(seen, p.sum(-1), 1.0)[..., None]
    dp = np.einsum("brhd,bchd->bhrc", g.astype(np.float64), vr)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True))
    dq = scale * np.einsum("bhrc,bchd->brhd", ds, kr)
    dk = scale * np.einsum("bhrc,brhd->bchd", ds, q.astype(np.float64)).reshape(B, Cn, Hk, ratio, D).sum(3)
    dv = np.einsum("bhrc,brhd->bchd", p, g.astype(np.float64)).reshape(B, Cn, Hk, ratio, Dv).sum(3)
    for a, w in zip(gb, (dq, dk, dv)):
        close(a, w)
    if mask is None:  # (the reference's CPU backward leaves the mask out, cpu_ref.c:396-406: compared only without one)
        r2, wb = exec_on(ref_lib, nnc.CPU_MEMORY, bcmd, nnc.NO_HINT, 0, bins, outs, backend=nnc.BACKEND_CPU_REF)
        assert r2 == 0
        for a, b in zip(gb, wb):
            close(a, b)


def test_attention_single_head_3d_and_head_projection(backend, ref_lib):
    rng = np.random.default_rng(6)
    B, R, Cn, D = 2, 21, 17, 32
    q, k, v = [(rng.random(s, dtype=F) - F(0.5)) for s in ((B, R, D), (B, Cn, D), (B, Cn, D))]
    fcmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD", 0.2, False)
    r1, got = exec_on(backend, nnc.GPU_MEMORY, fcmd, nnc.NO_HINT, 0, [q, k, v], [np.zeros((B, R, D), F)])
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, nnc.NO_HINT, 0, [q, k, v], [np.zeros((B, R, D), F)], backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    close(got[0], want[0])
    # unify heads: d = o (as [B, R, H * Dv]) w^T + bias
    H, Dv = 4, 8
    q4, k4, v4 = [(rng.random(s, dtype=F) - F(0.5)) for s in ((B, R, H, 16), (B, Cn, H, 16), (B, Cn, H, Dv))]
    w, bias = (rng.random((H * Dv, H * Dv), dtype=F) - F(0.5)), (rng.random(H * Dv, dtype=F) - F(0.5))
    outs = [np.zeros((B, R, H * Dv), F), np.zeros((B, H, R), F), np.zeros((B, R, H, Dv), F)]
    r1, got = exec_on(backend, nnc.GPU_MEMORY, fcmd, nnc.NO_HINT, 0, [q4, k4, v4, None, w, bias], outs)
    r2, want = exec_on(ref_lib, nnc.CPU_MEMORY, fcmd, nnc.NO_HINT, 0, [q4, k4, v4, None, w, bias], outs, backend=nnc.BACKEND_CPU_REF)
    assert r1 == 0 and r2 == 0
    close(got[2], want[2])
    close(got[0], want[0])


def test_attention_forward_on_the_matrix_cores_agrees_with_the_valu_kernel(backend, ref_lib):
    """The same forward through both kernels of cmd_attention.cpp (tuning key SDPA_MFMA): both within tolerance of the oracle, the matrix-core one recorded as such."""
    B, R, Cn, Hq, Hk, D, Dv = 2, 150, 70, 4, 2, 64, 64
    rng = np.random.default_rng(11)
    q = rng.random((B, R, Hq, D), dtype=F) - F(0.5)
    k = rng.random((B, Cn, Hk, D), dtype=F) - F(0.5)
    v = rng.random((B, Cn, Hk, Dv), dtype=F) - F(0.5)
    cmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD", float(1.0 / np.sqrt(D)), True)
    r0, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [q, k, v], [np.zeros((B, R, Hq, Dv), F)], backend=nnc.BACKEND_CPU_REF)
    assert r0 == 0
    seen = {}
    for mode in (1, 0):
        backend.tune_set("SDPA_MFMA", mode)
        backend.profile_enable(1)
        try:
            r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [q, k, v], [np.zeros((B, R, Hq, Dv), F)])
            backend.stream_wait(None)
            seen[mode] = [r[0] for r in backend.profile_records()]
        finally:
            backend.profile_enable(0)
            backend.tune_set("SDPA_MFMA", 1)
        assert r1 == 0
        close(got[0], want[0])
    assert any("sdpa_forw_mfma_kernel" in n for n in seen[1]) and not any("sdpa_forw_mfma_kernel" in n for n in seen[0]), seen


def test_attention_backward_on_the_matrix_cores_agrees_with_the_valu_kernels(backend, ref_lib):
    """dq and dk / dv through both kernel sets of cmd_attention.cpp (tuning key SDPA_MFMA): the same gradients within tolerance of each other and of the reference's
    CPU backward pass, the matrix-core kernels recorded as such (round 4: sdpa_dq_mfma_kernel, sdpa_dkv_mfma_kernel)."""
    B, R, Cn, Hq, Hk, D, Dv = 2, 150, 170, 4, 2, 64, 64
    rng = np.random.default_rng(12)
    q = rng.random((B, R, Hq, D), dtype=F) - F(0.5)
    k = rng.random((B, Cn, Hk, D), dtype=F) - F(0.5)
    v = rng.random((B, Cn, Hk, Dv), dtype=F) - F(0.5)
    g = rng.random((B, R, Hq, Dv), dtype=F) - F(0.5)
    cmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_BACKWARD", float(1.0 / np.sqrt(D)), True)
    ins = [g, None, None, q, k, v]
    outs = [np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)]
    r0, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, ins, outs, backend=nnc.BACKEND_CPU_REF)
    assert r0 == 0
    seen, res = {}, {}
    for mode in (1, 0):
        backend.tune_set("SDPA_MFMA", mode)
        backend.profile_enable(1)
        try:
            r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, ins, outs)
            backend.stream_wait(None)
            seen[mode] = [r[0] for r in backend.profile_records()]
        finally:
            backend.profile_enable(0)
            backend.tune_set("SDPA_MFMA", 1)
        assert r1 == 0
        res[mode] = got
        for a, b in zip(got, want):
            close(a, b)
    for a, b in zip(res[1], res[0]):
        close(a, b, tol=2e-5)
    assert any("sdpa_dq_mfma_kernel" in n for n in seen[1]) and any("sdpa_dkv_mfma_kernel" in n for n in seen[1]), seen
    assert not any("_mfma_kernel" in n for n in seen[0]), seen


HALF_CASES = [
    # B, R, C, Hq, Hk, D, Dv, causal, mask
    (2, 150, 70, 4, 2, 64, 64, True, None),      # the reference's flash_attn rows' kind of shape: ragged row / key blocks, grouped-query heads
    (1, 131, 97, 2, 2, 128, 96, False, None),
    (2, 40, 200, 3, 1, 48, 32, True, (2, 1)),    # an additive mask per batch item (CCV_16F like the rest), and causal
    (1, 33, 37, 2, 2, 16, 128, False, (1, 2)),
    (2, 20, 37, 4, 4, 24, 24, False, None),      # D not in 16s: the half tensors go through fp32 images and the fp32 kernels
]


@pytest.mark.parametrize("case", HALF_CASES, ids=[str(c) for c in HALF_CASES])
def test_attention_forward_in_half_precision(backend, ref_lib, case):
    """CCV_16F q / k / v / o (what the reference's flash_attn backend takes, cublas.tests.c:2752-2833, tolerance 3e-3 there): the f16 matrix-core kernel where its
    conditions hold (recorded as sdpa_forw_f16_kernel), fp32 images otherwise; against the oracle on the same half values in fp32."""
    B, R, Cn, Hq, Hk, D, Dv, causal, mask_shape = case
    H = np.float16
    rng = np.random.default_rng(21)
    q = (rng.random((B, R, Hq, D), dtype=F) - F(0.5)).astype(H)
    k = (rng.random((B, Cn, Hk, D), dtype=F) - F(0.5)).astype(H)
    v = (rng.random((B, Cn, Hk, Dv), dtype=F) - F(0.5)).astype(H)
    mask = None if mask_shape is None else ((rng.random(mask_shape + (R, Cn), dtype=F) - F(0.5)) * F(2)).astype(H)
    cmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD", float(1.0 / np.sqrt(D)), causal)
    extra, extra32 = ([mask], [mask.astype(F)]) if mask is not None else ([], [])
    r0, want = exec_on(ref_lib, nnc.CPU_MEMORY, cmd, nnc.NO_HINT, 0, [q.astype(F), k.astype(F), v.astype(F)] + extra32, [np.zeros((B, R, Hq, Dv), F)], backend=nnc.BACKEND_CPU_REF)
    assert r0 == 0
    backend.profile_enable(1)
    try:
        r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [q, k, v] + extra, [np.zeros((B, R, Hq, Dv), H), np.zeros((B, Hq, R), F)])
        backend.stream_wait(None)
        names = [r[0] for r in backend.profile_records()]
    finally:
        backend.profile_enable(0)
    assert r1 == 0
    native = D % 16 == 0 and Dv % 32 == 0
    assert any("sdpa_forw_f16_kernel" in n for n in names) == native, names
    np.testing.assert_allclose(got[0].astype(F), want[0], rtol=0, atol=3e-3)
    if native:  # the log-sum-exp rows (fp32) of the rows that see a key
        s = np.einsum("brhd,bchd->bhrc", q.astype(np.float64), np.repeat(k, Hq // Hk, axis=2).astype(np.float64)) / np.sqrt(D)
        if mask is not None:
            s = s + mask.astype(np.float64)
        if causal:
            vis = np.arange(R)[:, None] - R + Cn + 1
            s = np.where(np.arange(Cn)[None, :] < vis, s, -np.inf)
        seen = np.isfinite(s).any(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            lse = np.log(np.exp(s - s.max(-1, keepdims=True).clip(-1e30)).sum(-1)) + s.max(-1).clip(-1e30)
        np.testing.assert_allclose(got[1][seen], lse[seen], rtol=2e-3, atol=2e-3)


HALF_BACK_CASES = [
    # B, R, C, Hq, Hk, D, Dv, causal, mask
    (2, 150, 170, 4, 2, 64, 64, True, None),
    (1, 131, 97, 2, 2, 128, 96, False, (1, 2)),
    (2, 70, 40, 3, 1, 32, 32, True, (2, 3)),
    (1, 33, 37, 2, 2, 48, 32, False, None),      # D not in 32s: fp32 images and the fp32 kernels
]


@pytest.mark.parametrize("case", HALF_BACK_CASES, ids=[str(c) for c in HALF_BACK_CASES])
def test_attention_backward_in_half_precision(backend, ref_lib, case):
    """CCV_16F g / q / k / v -> dq / dk / dv (the reference's flash_attn backward, cublas.tests.c:2835-): the f16 matrix-core kernels where their conditions hold
    (sdpa_dq_f16_kernel, sdpa_dkv_f16_kernel, behind the half-precision forward re-run), fp32 images otherwise; against the closed-form float64 gradients of the
    same half values."""
    B, R, Cn, Hq, Hk, D, Dv, causal, mask_shape = case
    H = np.float16
    rng = np.random.default_rng(31)
    q = (rng.random((B, R, Hq, D), dtype=F) - F(0.5)).astype(H)
    k = (rng.random((B, Cn, Hk, D), dtype=F) - F(0.5)).astype(H)
    v = (rng.random((B, Cn, Hk, Dv), dtype=F) - F(0.5)).astype(H)
    g = (rng.random((B, R, Hq, Dv), dtype=F) - F(0.5)).astype(H)
    mask = None if mask_shape is None else ((rng.random(mask_shape + (R, Cn), dtype=F) - F(0.5)) * F(2)).astype(H)
    scale = float(1.0 / np.sqrt(D))
    cmd = sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_BACKWARD", scale, causal)
    backend.profile_enable(1)
    try:
        r1, got = exec_on(backend, nnc.GPU_MEMORY, cmd, nnc.NO_HINT, 0, [g, None, None, q, k, v] + ([mask] if mask is not None else []), [np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)])
        backend.stream_wait(None)
        names = [r[0] for r in backend.profile_records()]
    finally:
        backend.profile_enable(0)
    assert r1 == 0
    native = D % 32 == 0 and Dv % 32 == 0
    assert (any("sdpa_dq_f16_kernel" in n for n in names) and any("sdpa_dkv_f16_kernel" in n for n in names)) == native, names
    ratio = Hq // Hk
    q64, g64 = q.astype(np.float64), g.astype(np.float64)
    kr, vr = np.repeat(k, ratio, axis=2).astype(np.float64), np.repeat(v, ratio, axis=2).astype(np.float64)
    s = np.einsum("brhd,bchd->bhrc", q64, kr) * scale
    if mask is not None:
        s = s + mask.astype(np.float64)
    if causal:
        vis = np.arange(R)[:, None] - R + Cn + 1
        s = np.where(np.arange(Cn)[None, :] < vis, s, -np.inf)
    seen = np.isfinite(s).any(-1)
    with np.errstate(invalid="ignore"):
        p = np.exp(s - np.where(seen, s.max(-1), 0.0)[..., None])
    p = np.where(np.isfinite(s), p, 0.0)
    p = p / np.where(seen, p.sum(-1), 1.0)[..., None]
    dp = np.einsum("brhd,bchd->bhrc", g64, vr)
    ds = p * (dp - (p * dp).sum(-1, keepdims=True))
    dq = scale * np.einsum("bhrc,bchd->brhd", ds, kr)
    dk = scale * np.einsum("bhrc,brhd->bchd", ds, q64).reshape(B, Cn, Hk, ratio, D).sum(3)
    dv = np.einsum("bhrc,brhd->bchd", p, g64).reshape(B, Cn, Hk, ratio, Dv).sum(3)
    for a, w in zip(got, (dq, dk, dv)):
        close(a.astype(F), w, tol=5e-3)
