"""CCV_NNC_LSTM_FORWARD / BACKWARD (ccv_amd/csrc/cmd_lstm.cpp; replaces lib/nnc/cmd/rnn/gpu/ccv_nnc_lstm_gpu_cudnn.cu) against oracle/lstm_numpy.py.

The reference has no CPU LSTM and its tests (test/int/nnc/lstm.tests.c) assert nothing, so the oracle restates cuDNN's published LSTM in float64
("parity unpinned", see its header); what pins the ORACLE is the first test here: its gradients equal central differences of its forward pass.
The cases walk the eight configurations of the reference's tests (layers, no initial / final states, dropout, projection, projection + both
directions, gradients with and without state gradients) at sizes the emulator finishes, plus batch-first tensors, per-item sequence lengths,
no bias, more than one workgroup tile each way, and CCV_16F tensors.  Tolerance: 1e-4 of the tensor's largest value in fp32 (north_star's fp32
bound), 2e-2 in half precision (x, w, y and the gradients are half tensors; the tape between the two commands stays fp32 inside the reserved space)."""
import os
import sys
import numpy as np
import pytest
from ccv_amd import nnc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import lstm_numpy as oracle  # noqa: E402

F = np.float32


def close(got, want, tol=1e-4):
    bound = tol * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max())
    assert err <= bound, "max |diff| %.3g > %.3g" % (err, bound)


def test_oracle_gradients_match_central_differences():
    rng = np.random.default_rng(3)
    T, B, I, H, P, L, D = 4, 3, 5, 6, 4, 2, 2
    n = oracle.weight_count(I, H, P, L, D, True)
    x = rng.standard_normal((T, B, I)) * 0.5
    w = rng.standard_normal(n) * 0.3
    hx = rng.standard_normal((L * D, B, P)) * 0.5
    cx = rng.standard_normal((L * D, B, H)) * 0.5
    lens = np.array([4, 2, 3])
    masks = [np.where(rng.random((T, B, D * P)) < 0.3, 0.0, 1 / 0.7)]
    gy, ghy, gcy = rng.standard_normal((T, B, D * P)), rng.standard_normal((L * D, B, P)), rng.standard_normal((L * D, B, H))

    def loss(x_, w_, hx_, cx_):
        y, hy, cy, _ = oracle.forward(x_, w_, H, P, L, True, True, hx_, cx_, lens, masks)
        return float((y * gy).sum() + (hy * ghy).sum() + (cy * gcy).sum())

    _, _, _, tape = oracle.forward(x, w, H, P, L, True, True, hx, cx, lens, masks)
    dx, dhx, dcx, dw = oracle.backward(gy, tape, ghy, gcy)
    eps = 1e-6
    for name, arr, grad, args in (("x", x, dx, 0), ("w", w, dw, 1), ("hx", hx, dhx, 2), ("cx", cx, dcx, 3)):
        flat = arr.reshape(-1)
        for k in rng.choice(flat.size, size=min(40, flat.size), replace=False):
            keep = flat[k]
            vals = []
            for sgn in (1, -1):
                flat[k] = keep + sgn * eps
                vals.append(loss(x, w, hx, cx))
            flat[k] = keep
            num = (vals[0] - vals[1]) / (2 * eps)
            assert abs(num - grad.reshape(-1)[k]) <= 1e-6 * max(1.0, abs(num)), (name, k, num, grad.reshape(-1)[k])


#        T,  B,  I,  H,  P, L, bias, batch_first, bidirectional, lens,          dropout, states, two_d
CASES = [
    (5,  1, 24, 24,  0, 3, 1, 0, 0, None,                  0.0, True,  False),   # "LSTM forward" / "LSTM backward" (lstm.tests.c:36, :281), three layers here
    (5,  1, 24, 24,  0, 2, 1, 0, 0, None,                  0.0, False, True),    # "... without hx, cx, hy, cy" (:89, :441): x is [T, I]
    (4,  3,  6, 16,  0, 3, 1, 0, 0, None,                  0.4, True,  False),   # "... with dropout" (:122)
    (5,  2,  9, 16,  8, 2, 1, 0, 0, None,                  0.0, True,  False),   # "... with projection" (:175)
    (4,  3,  6, 12,  4, 2, 1, 0, 1, None,                  0.0, True,  False),   # "... with projection, bidirectional" (:228)
    (4,  5,  7, 12,  0, 2, 1, 0, 1, None,                  0.0, True,  False),   # both directions, no projection
    (6,  4,  8, 16,  0, 2, 1, 1, 1, [6, 3, 1, 4],          0.0, True,  False),   # batch-first tensors, an own length per item
    (5,  3,  6, 12,  8, 2, 1, 1, 1, [2, 5, 4],             0.3, True,  False),   # everything at once
    (3,  2,  5,  8,  0, 1, 0, 0, 0, None,                  0.0, True,  False),   # no bias
    (3, 20, 70, 80,  0, 1, 1, 0, 0, [3, 1, 2] * 6 + [3, 2], 0.0, True, False),   # two tiles of hidden units, of batch rows and of the recurrent reduction
    (3,  3, 10, 144, 0, 1, 1, 0, 1, [3, 2, 1],             0.0, True,  False),   # hidden size > 128: two register chunks per lane in the one-launch kernels
    (2,  2,  5, 272, 0, 1, 1, 0, 0, None,                  0.0, True,  False),   # > 256: the widest forward kernel, three chunks backward
    (4,  5, 12, 128, 0, 2, 1, 0, 1, [4, 1, 3, 2, 4],       0.0, True,  False),   # the rows form at its widest (512 threads, R = 128 registers per thread), an odd batch, both directions
    (3,  7,  9, 100, 0, 2, 1, 1, 0, [3, 3, 1, 2, 3, 1, 2], 0.3, True,  False),   # ... a hidden size between its register tiers (96 < 100 <= 128), batch-first
]


def _run(lib, case, dtype=F, tol=1e-4, persistent=2):
    # 2: the defaults (hidden size <= 128 without projection: a workgroup per batch row -- per two rows when the hidden size is no multiple of four --, R in its
    # registers); 3: the two-row form of that for every size; 1: hidden units split over workgroups that hand the state to each other; 0: a launch per step
    lib.tune_set("LSTM_PERSISTENT", 1 if persistent else 0)
    lib.tune_set("LSTM_ROWS", {2: 1, 3: 2}.get(persistent, 0))
    try:
        _run_inner(lib, case, dtype, tol, persistent)
    finally:
        lib.tune_set("LSTM_PERSISTENT", 1)
        lib.tune_set("LSTM_ROWS", 1)


def _run_inner(lib, case, dtype, tol, persistent):
    T, B, I, H, P, L, bias, batch_first, bidir, lens, dropout, states, two_d = case
    D = 2 if bidir else 1
    Pe = P or H
    rng = np.random.default_rng(11)
    nw = oracle.weight_count(I, H, Pe, L, D, bias)
    x = (rng.random((T, B, I), dtype=F) - F(0.5)).astype(dtype)
    w = ((rng.random(nw, dtype=F) - F(0.5)) * F(0.6)).astype(dtype)
    hx = (rng.random((L * D, B, Pe), dtype=F) - F(0.5)).astype(dtype) if states else None
    cx = (rng.random((L * D, B, H), dtype=F) - F(0.5)).astype(dtype) if states else None
    fcmd = nnc.CMD_LSTM_FORWARD(H, P, L, bias, batch_first, bidir, dropout, 0)
    dt = nnc._NP_DT[np.dtype(dtype)]
    rbytes = lib.dll.nnc_mi355x_lstm_reserve_space_size(fcmd, dt, I, B, T)
    rrows = (rbytes // np.dtype(dtype).itemsize + H - 1) // H
    assert rrows > 0

    def lay(a):  # the tensor the host would hand over for the sequence-major array a
        if two_d:
            return np.ascontiguousarray(a[:, 0])
        return np.ascontiguousarray(a.transpose(1, 0, 2)) if batch_first else a

    def unlay(a):
        if two_d:
            return a[:, None]
        return a.transpose(1, 0, 2) if batch_first else a

    def gpu(a):
        return None if a is None else lib.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, dt, a.shape), np.ascontiguousarray(a))

    xs_t = None if lens is None else lib.tensor(nnc.tensor_param(nnc.CPU_MEMORY, nnc.NHWC, nnc.CCV_32S, (B,)), np.asarray(lens, np.int32))
    w2 = w.reshape(-1, 1) if nw % H else w.reshape(-1, H)
    x_t, hx_t, cx_t, w_t = gpu(lay(x)), gpu(hx), gpu(cx), gpu(w2)
    y_t = gpu(np.zeros_like(lay(np.zeros((T, B, D * Pe), dtype))))
    hy_t = gpu(np.zeros((L * D, B, Pe), dtype)) if states else None
    cy_t = gpu(np.zeros((L * D, B, H), dtype)) if states else None
    r_t = gpu(np.full((rrows, H), np.nan, dtype))
    assert lib.cmd_exec(fcmd, nnc.NO_HINT, 0, [x_t, xs_t, hx_t, cx_t, w_t], [y_t, hy_t, cy_t, r_t]) == 0
    if dtype == F:  # which forward ran: the whole sequence in one launch, or a launch per step (projection: always per step)
        one_launch = persistent and not (P and P != H)
        assert lib.dll.nnc_mi355x_last_kernel_name().decode() == (("lstm_rows_forw" if persistent >= 2 and H <= 128 else "lstm_seq_forw") if one_launch else "lstm_step_forw")
    # (the reserved space holds fp32 planes whatever the command's data type: a CCV_16F tensor is only the container the host sized)
    r = np.ascontiguousarray(r_t.numpy()).reshape(-1).view(np.float32).astype(np.float64)
    masks = None
    if dropout > 0:  # the scales the command drew: plane S - 1 of every (pseudo-layer, step) of the reserved space (cmd_lstm.cpp's header)
        S = 5 + (1 if Pe != H else 0) + 1
        planes = r[:L * D * T * S * B * H].reshape(L * D, T, S, B, H)[:, :, S - 1, :, :Pe]
        masks = []
        for l in range(L - 1):
            m = np.zeros((T, B, D * Pe))
            for d in range(D):
                pl = planes[l * D + d]
                m[:, :, d * Pe:(d + 1) * Pe] = pl[::-1] if d else pl
            keep = 1.0 / (1.0 - dropout)
            assert np.all((m == 0) | (np.abs(m - keep) < 1e-6))
            masks.append(m)
        drawn = np.concatenate([m.ravel() for m in masks])
        assert 0.05 < (drawn == 0).mean() < 0.8
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    y, hy, cy, tape = oracle.forward(x64, w64, H, Pe, L, bool(bias), bool(bidir), hx, cx, lens, masks)
    close(unlay(y_t.numpy()), y, tol)
    if states:
        close(hy_t.numpy(), hy, tol)
        close(cy_t.numpy(), cy, tol)
    # the same command in test mode: no reserved space, the same numbers (no dropout there)
    if dropout == 0:
        icmd = nnc.CMD_LSTM_FORWARD(H, P, L, bias, batch_first, bidir, dropout, 1)
        assert lib.dll.nnc_mi355x_lstm_reserve_space_size(icmd, dt, I, B, T) == 0
        y2_t = gpu(np.zeros_like(lay(np.zeros((T, B, D * Pe), dtype))))
        assert lib.cmd_exec(icmd, nnc.NO_HINT, 0, [x_t, xs_t, hx_t, cx_t, w_t], [y2_t, None, None]) == 0
        np.testing.assert_array_equal(y2_t.numpy(), y_t.numpy())
    # backward
    gy = (rng.random((T, B, D * Pe), dtype=F) - F(0.5)).astype(dtype)
    ghy = (rng.random((L * D, B, Pe), dtype=F) - F(0.5)).astype(dtype) if states else None
    gcy = (rng.random((L * D, B, H), dtype=F) - F(0.5)).astype(dtype) if states else None
    bcmd = nnc.CMD_LSTM_BACKWARD(H, P, L, bias, batch_first, bidir, dropout, 0)
    dx_t, dw_t = gpu(np.zeros_like(lay(x))), gpu(np.full_like(w2, np.nan))
    dhx_t = gpu(np.zeros((L * D, B, Pe), dtype)) if states else None
    dcx_t = gpu(np.zeros((L * D, B, H), dtype)) if states else None
    ins = [gpu(lay(gy)), gpu(ghy), gpu(gcy), None, x_t, xs_t, hx_t, cx_t, w_t, y_t, hy_t, cy_t, r_t]
    assert lib.cmd_exec(bcmd, nnc.NO_HINT, 0, ins, [dx_t, None, dhx_t, dcx_t, dw_t]) == 0
    if dtype == F:
        assert lib.dll.nnc_mi355x_last_kernel_name().decode() == (("lstm_rows_back" if persistent >= 2 and H <= 128 else "lstm_seq_back") if persistent and not (P and P != H) else "lstm_step_back")
    dx, dhx, dcx, dw = oracle.backward(gy.astype(np.float64), tape, ghy, gcy)
    close(unlay(dx_t.numpy()), dx, tol)
    close(dw_t.numpy().reshape(-1), dw, tol)
    if states:
        close(dhx_t.numpy(), dhx, tol)
        close(dcx_t.numpy(), dcx, tol)


@pytest.mark.parametrize("persistent", [2, 3, 1, 0], ids=["rows", "two-rows", "one-launch", "per-step"])
@pytest.mark.parametrize("case", CASES, ids=[str(c[:9]) + ("+lens" if c[9] else "") + ("+drop" if c[10] else "") for c in CASES])
def test_lstm_forward_backward(backend, case, persistent):
    """The three forms of the sweep (tuning keys LSTM_PERSISTENT, LSTM_ROWS): the whole sequence of a pseudo-layer in one launch with a workgroup per two batch rows
    and all of R in its registers (hidden size <= 128, nothing passes between workgroups); in one launch with the hidden units split over workgroups that hand the state
    to each other through tagged words; and a launch per step.  The backward command reads the reserved space any of them wrote."""
    if persistent >= 2 and case[3] > 128:
        pytest.skip("hidden size > 128: the default is the split form (the next parameter)")
    if persistent and case[4] and case[4] != case[3]:
        pytest.skip("a projection always takes the per-step form")
    _run(backend, case, persistent=persistent)


def test_lstm_in_half_precision(backend):
    """CCV_16F tensors (the reference's row lists CCV_16F, ccv_nnc_lstm_gpu_cudnn.cu:255): fp32 arithmetic on fp32 images of the half tensors."""
    _run(backend, (4, 3, 8, 16, 0, 2, 1, 0, 1, [4, 2, 3], 0.0, True, False), dtype=np.float16, tol=2e-2)


def test_reserved_space_fits_what_the_reference_tests_allocate(backend):
    """test/int/nnc/lstm.tests.c sizes the reserved-space tensor with its own r_dim() (:23-34, cuDNN's need) instead of asking the command: the eight
    configurations it runs must fit."""
    def r_dim(bidir, dropout, B, L, T, H, P):
        D = 2 if bidir else 1
        k = 5 if H == P else 6
        return D * B * ((k + (1 if dropout else 0)) * L * T + 2 * L * (T - 1))
    for H, P, L, bidir, dropout in ((24, 0, 6, 0, 0.0), (24, 0, 6, 0, 0.5), (24, 12, 6, 0, 0.0), (24, 12, 6, 1, 0.0)):
        cmd = nnc.CMD_LSTM_FORWARD(H, P, L, 1, 0, bidir, dropout, 0)
        need = backend.dll.nnc_mi355x_lstm_reserve_space_size(cmd, nnc.CCV_32F, 24, 1, 5)
        assert 0 < need <= 4 * 24 * r_dim(bidir, dropout, 1, L, 5, H, P or H), (H, P, L, bidir, dropout, need)



TORCH_CASES = 4  # tests/lstm_torch_check.py's table


@pytest.mark.parametrize("case", range(TORCH_CASES))
def test_oracle_matches_an_independent_lstm(case):
    """oracle/lstm_numpy.py against torch.nn.LSTM on the CPU (float64), see tests/lstm_torch_check.py -- in a process of its own: torch brings its own OpenMP runtime,
    which must not meet the one the reference's library (loaded by the rest of this tier) is linked to."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lstm_torch_check.py"), str(case)], capture_output=True, text=True, timeout=600)
    if r.returncode == 77:
        pytest.skip("torch is not importable here")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
