"""Batch norm (training) on the CLUSTER kernels (ccv_amd/csrc/cmd_norm.cpp, round 4): a cluster of workgroups per channel holds the channel in registers
between the statistics and the apply pass, the workgroups of a cluster hand their partial sums to each other inside the launch.  Against the reference's CPU
batch norm (lib/nnc/cmd/norm/ccv_nnc_batch_norm_cpu_ref.c:44-232 forward, :300-470 backward) on [N][C][H][W] tensors, fp32 and CCV_16F, with the chunks per
workgroup forced down (BN_CLUSTER = chunks) so that small tensors take several workgroups per channel -- ragged shares, ragged last workgroups, the ReLU bit.
CPU tier: the kernel sources on the emulator, whose concurrent launch keeps a WINDOW of workgroups resident and can dispatch them in reverse or shuffled order
(HIP promises none): the hand-over protocol must not care."""
import os
import subprocess
import sys
import numpy as np
import pytest
from ccv_amd import nnc
from harness import make_tensors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, H = np.float32, np.float16


def _run(lib, mem, x, g, scale, bias, mean, var, sshape, backend_id=None, relu=False):
    C = x.shape[1]
    axes = (0, 2, 3)
    r = lambda a: a.reshape(sshape).copy()
    tx, tg = make_tensors(lib, mem, [x, g], "NCHW")
    ts = make_tensors(lib, mem, [r(scale), r(bias), r(mean), r(var)], "NCHW")
    ty, th = make_tensors(lib, mem, [np.zeros_like(x), np.zeros_like(x)], "NCHW")
    tsm, tsi, tds, tdb = make_tensors(lib, mem, [np.zeros(sshape, F) for _ in range(4)], "NCHW")
    c = nnc.CMD_BATCH_NORM_FORWARD(1e-4, 0, 0.9, *axes)
    cb = nnc.CMD_BATCH_NORM_BACKWARD(1e-4, 0, 0.9, *axes)
    if backend_id is not None:
        c.backend = backend_id; cb.backend = backend_id
    if relu:
        c.algorithm = nnc.BNORM_ALGO_FUSE_RELU
    assert lib.cmd_exec(c, nnc.NO_HINT, 0, [tx] + ts, [ty, ts[2], ts[3], tsm, tsi]) == 0
    assert lib.cmd_exec(cb, nnc.NO_HINT, 0, [tg] + [None] * 4 + [tx, ts[0]] + [None] * 6 + [tsm, tsi], [th, tds, tdb]) == 0
    return [t.numpy() for t in (ty, th, ts[2], ts[3], tsm, tsi, tds, tdb)]


CASES = [  # (N, C, H, W), chunks per workgroup (0 = the kernels' own capacity: one workgroup per channel at these sizes)
    ((4, 6, 8, 8), 0), ((4, 6, 8, 8), 16), ((3, 5, 6, 10), 7), ((5, 3, 12, 12), 33), ((2, 4, 40, 44), 100), ((7, 2, 4, 4), 2), ((6, 5, 14, 14), 64),
    ((5, 6, 7, 7), 0), ((5, 6, 7, 7), 60), ((3, 4, 5, 6), 11),  # planes that are not whole 16-byte chunks: 8-byte / 4-byte chunks (7 x 7 fp32: 196 bytes, 14 x 14 halves: 392)
]


@pytest.mark.parametrize("shape,cap", CASES, ids=["%dx%dx%dx%d-cap%d" % (s + (c,)) for s, c in CASES])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("relu", [False, True], ids=["plain", "relu"])
def test_batch_norm_cluster_kernels(backend, ref_lib, shape, cap, dtype, relu):
    rng = np.random.default_rng(31)
    T = F if dtype == "f32" else H
    C = shape[1]
    x = (2.0 * rng.standard_normal(shape) + 0.5).astype(T)
    g = rng.standard_normal(shape).astype(T)
    scale, bias = (rng.random(C, dtype=F) + F(1.0)), (rng.random(C, dtype=F) - F(0.5))
    mean, var = (rng.random(C, dtype=F) - F(0.5)), rng.random(C, dtype=F) + F(0.5)
    if shape[2] * shape[3] * np.dtype(T).itemsize % 4 != 0:
        pytest.skip("planes of whole dwords only (7 x 7 halves: the plane kernels)")
    old = backend.tune_get("BN_CLUSTER")
    n0 = backend.dll.nnc_mi355x_debug_bn_cluster_launches()
    try:
        if cap:
            backend.tune_set("BN_CLUSTER", cap)
        got = _run(backend, nnc.GPU_MEMORY, x, g, scale, bias, mean, var, (C,), relu=relu)
    finally:
        backend.tune_set("BN_CLUSTER", old)
    assert backend.dll.nnc_mi355x_debug_bn_cluster_launches() == n0 + 2  # forward and backward both took the cluster kernels
    want = _run(ref_lib, nnc.CPU_MEMORY, x.astype(F), g.astype(F), scale, bias, mean, var, (1, C, 1, 1), nnc.BACKEND_CPU_REF)
    if relu:
        want[0] = np.maximum(want[0], 0)
    ytol = dict(rtol=2e-4, atol=2e-5) if T is F else dict(rtol=2e-3, atol=2e-3 * max(1.0, float(np.abs(want[0]).max())))
    np.testing.assert_allclose(got[0].astype(F), want[0], err_msg="y", **ytol)
    np.testing.assert_allclose(got[1].astype(F), want[1], err_msg="h", **(ytol if T is F else dict(rtol=2e-3, atol=2e-3 * max(1.0, float(np.abs(want[1]).max())))))
    for a, b, what in zip(got[2:], want[2:], ("mean", "var", "saved_mean", "saved_inv_std", "dscale", "dbias")):
        np.testing.assert_allclose(a.reshape(-1), b.reshape(-1), rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(b).max())), err_msg=what)


def test_cluster_kernels_agree_with_the_plane_kernels_bit_for_bit_in_the_apply_pass(backend):
    """Same statistics scheme (share sums, centred second moments, Chan's fold), another partition: y / h agree to a few ulp and the choice is a switch."""
    rng = np.random.default_rng(32)
    shape = (6, 8, 12, 12)
    x, g = (rng.standard_normal(shape) * 3).astype(F), rng.standard_normal(shape).astype(F)
    C = shape[1]
    args = (rng.random(C, dtype=F) + F(1.0), rng.random(C, dtype=F), rng.random(C, dtype=F), rng.random(C, dtype=F) + F(0.5))
    old = backend.tune_get("BN_CLUSTER")
    try:
        backend.tune_set("BN_CLUSTER", 40)
        a = _run(backend, nnc.GPU_MEMORY, x, g, *args, (C,))
        backend.tune_set("BN_CLUSTER", 0)
        n0 = backend.dll.nnc_mi355x_debug_bn_cluster_launches()
        b = _run(backend, nnc.GPU_MEMORY, x, g, *args, (C,))
        assert backend.dll.nnc_mi355x_debug_bn_cluster_launches() == n0
    finally:
        backend.tune_set("BN_CLUSTER", old)
    for u, v in zip(a, b):
        np.testing.assert_allclose(u, v, rtol=2e-5, atol=2e-6)


ORDER_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from ccv_amd import nnc
import test_bn_cluster as t
L = nnc.load(%r)
rng = np.random.default_rng(33)
shape = (5, 4, 8, 12)
x, g = (rng.standard_normal(shape) * 2).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
C = shape[1]
args = (rng.random(C, dtype=np.float32) + 1, rng.random(C, dtype=np.float32), rng.random(C, dtype=np.float32), rng.random(C, dtype=np.float32) + 0.5)
L.tune_set("BN_CLUSTER", 9)    # 14 workgroups per channel, 56 per launch, more than the resident window
out = t._run(L, nnc.GPU_MEMORY, x, g, *args, (C,))
assert L.dll.nnc_mi355x_debug_bn_cluster_launches() == 2
np.savez(sys.argv[1], *out)
"""


def test_hand_over_does_not_depend_on_dispatch_order_or_residency(emu_lib, tmp_path):
    """The emulator dispatches the workgroups forward / in reverse / shuffled, 3 to 16 of them resident: every run gives the same bits (the tickets make a
    workgroup's place in its cluster independent of where the dispatcher started it; the folds run in a fixed order)."""
    from conftest import emu_so
    so = emu_so()
    outs = []
    for order, resident in (("forward", 16), ("reverse", 15), ("shuffle", 14), ("shuffle", 30)):
        f = tmp_path / ("%s_%d.npz" % (order, resident))
        env = dict(os.environ, NNC_EMU_DISPATCH_ORDER=order, NNC_EMU_RESIDENT_BLOCKS=str(resident))
        r = subprocess.run([sys.executable, "-c", ORDER_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), so), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(np.load(f))
    for o in outs[1:]:
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], o[k]), k
