"""COMM_ALLREDUCE / BROADCAST / REDUCE in the reference's single-process N-device form on the 4-device CPU emulator (its
in-process RCCL stand-in): collectives issued from TWO stream contexts (each gets its own communicator set, as the reference
keeps them per stream: lib/nnc/gpu/ccv_nnc_compat.cu:1415-1445), consecutive commands coalesced into one RCCL group, results
exact.  Runs in a subprocess (the device count of the emulator is fixed at load time)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import ctypes as C, sys, numpy as np
    sys.path.insert(0, %r)
    from ccv_amd import nnc
    L = nnc.load(%r)
    assert L.device_count() == 4
    rng = np.random.default_rng(3)
    F = nnc.CCV_32F
    def on(dev, arr):
        return L.tensor(nnc.GPU_TENSOR_NHWC(dev, F, *arr.shape), arr)
    sA, sB = L.stream_new(0), L.stream_new(0)
    xa = [rng.standard_normal(1000).astype(np.float32) for _ in range(4)]
    xb = [rng.standard_normal((8, 16)).astype(np.float32) for _ in range(4)]
    ta, tb = [on(d, xa[d]) for d in range(4)], [on(d, xb[d]) for d in range(4)]
    ar = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
    c0, g0 = C.c_long(), C.c_long()
    L.dll.nnc_mi355x_comm_stats(C.byref(c0), C.byref(g0))
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, ta, ta, sA) == 0          # two commands back to back, different stream contexts:
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, tb, tb, sB) == 0          # recorded, then issued together
    L.stream_wait(sA); L.stream_wait(sB)
    c1, g1 = C.c_long(), C.c_long()
    L.dll.nnc_mi355x_comm_stats(C.byref(c1), C.byref(g1))
    assert c1.value - c0.value == 8 and g1.value - g0.value == 1, (c1.value - c0.value, g1.value - g0.value)
    sa, sb = sum(x.astype(np.float64) for x in xa), sum(x.astype(np.float64) for x in xb)
    for d in range(4):
        np.testing.assert_allclose(ta[d].numpy(), sa, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tb[d].numpy(), sb, rtol=1e-6, atol=1e-6)
    # broadcast from device 2, reduce to device 1; a non-COMM command in between is ordered after the recorded collective
    src = on(2, xa[0]); outs = [on(d, np.zeros(1000, np.float32)) for d in range(4)]
    assert L.cmd_exec(nnc.generic_cmd("COMM_BROADCAST_FORWARD"), nnc.NO_HINT, 0, [src], outs, sA) == 0
    assert L.cmd_exec(nnc.CMD_SCALAR_MUL_FORWARD(2.0), nnc.NO_HINT, 0, [outs[0]], [outs[0]], sA) == 0
    L.stream_wait(sA)
    np.testing.assert_array_equal(outs[0].numpy(), 2 * xa[0])
    np.testing.assert_array_equal(outs[3].numpy(), xa[0])
    red = on(1, np.zeros(1000, np.float32))
    assert L.cmd_exec(nnc.generic_cmd("COMM_REDUCE_FORWARD"), nnc.NO_HINT, 0, [on(d, xa[d]) for d in range(4)], [red], sB) == 0
    L.stream_wait(sB)
    np.testing.assert_allclose(red.numpy(), sa, rtol=1e-6, atol=1e-6)
    L.stream_free(sA); L.stream_free(sB)   # releases the two contexts' communicator sets
    print("OK")
""")


def test_comm_two_stream_contexts_coalesced(emu_lib):
    env = dict(os.environ, NNC_EMU_DEVICE_COUNT="4")
    from conftest import emu_so
    so = emu_so()
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, so)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# CCV_16F rows: the reference registers CCV_32F | CCV_16F (lib/nnc/cmd/comm/gpu/ccv_nnc_comm_gpu_nccl.cu:65,76,173-206) and maps the element
# type with ccv_nnc_nccl_datatype (lib/nnc/gpu/ccv_nnc_compat.cu:1447-1460); the f16 trainers (bin/nnc/imagenet.c:344) all-reduce half gradients.
SCRIPT_HALF = textwrap.dedent("""
    import ctypes as C, sys, numpy as np
    sys.path.insert(0, %r)
    from ccv_amd import nnc
    L = nnc.load(%r)
    assert L.device_count() == 4
    rng = np.random.default_rng(5)
    H = nnc.CCV_16F
    def on(dev, arr, dt=H):
        return L.tensor(nnc.GPU_TENSOR_NHWC(dev, dt, *arr.shape), arr)
    s = L.stream_new(0)
    # small integers: every partial sum is exact in half precision, so the result is order-independent and must be exact
    xs = [rng.integers(-64, 64, 4096).astype(np.float16) for _ in range(4)]
    ts = [on(d, xs[d]) for d in range(4)]
    ar = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, ts, ts, s) == 0
    L.stream_wait(s)
    want = sum(x.astype(np.float32) for x in xs).astype(np.float16)
    for d in range(4):
        got = ts[d].numpy()
        assert got.dtype == np.float16
        np.testing.assert_array_equal(got, want)
    # general values, two devices: one rounding of an exact two-term sum
    ya = [rng.standard_normal(1000).astype(np.float16) for _ in range(2)]
    ty = [on(d, ya[d]) for d in range(2)]
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, ty, ty, s) == 0
    L.stream_wait(s)
    want2 = (ya[0].astype(np.float32) + ya[1].astype(np.float32)).astype(np.float16)
    for d in range(2):
        np.testing.assert_array_equal(ty[d].numpy(), want2)
    # broadcast / reduce in half
    src = on(3, xs[0]); outs = [on(d, np.zeros(4096, np.float16)) for d in range(4)]
    assert L.cmd_exec(nnc.generic_cmd("COMM_BROADCAST_FORWARD"), nnc.NO_HINT, 0, [src], outs, s) == 0
    red = on(1, np.zeros(4096, np.float16))
    assert L.cmd_exec(nnc.generic_cmd("COMM_REDUCE_FORWARD"), nnc.NO_HINT, 0, [on(d, xs[d]) for d in range(4)], [red], s) == 0
    L.stream_wait(s)
    for d in range(4):
        np.testing.assert_array_equal(outs[d].numpy(), xs[0])
    np.testing.assert_array_equal(red.numpy(), want)
    # a command mixing element types is refused, not reinterpreted
    mixed = [on(0, xs[0]), on(1, xs[1].astype(np.float32), nnc.CCV_32F)]
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, mixed, mixed, s) != 0
    L.stream_free(s)
    print("OK")
""")


def test_comm_half_precision_rows(emu_lib):
    env = dict(os.environ, NNC_EMU_DEVICE_COUNT="4")
    from conftest import emu_so
    so = emu_so()
    r = subprocess.run([sys.executable, "-c", SCRIPT_HALF % (ROOT, so)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


LOCK_ORDER_SCRIPT = textwrap.dedent("""
    import ctypes as C, sys, threading, time, numpy as np
    sys.path.insert(0, %r)
    from ccv_amd import nnc
    L = nnc.load(%r)
    assert L.device_count() >= 2
    rng = np.random.default_rng(5)
    F = nnc.CCV_32F
    def on(dev, arr):
        return L.tensor(nnc.GPU_TENSOR_NHWC(dev, F, *arr.shape), arr)
    n, h, w, c, k = 2, 9, 10, 16, 32
    a = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((k, 3, 3, c)) / (9 * c)).astype(np.float32)
    b = (0.05 * rng.standard_normal(k)).astype(np.float32)
    s1 = L.stream_new(1)                         # the stream of device 1
    ins = [on(1, a), on(1, wt), on(1, b)]
    out = on(1, np.zeros((n, h, w, k), np.float32))
    cmd, hint = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), nnc.HINT((1, 1), (1, 1))
    assert L.cmd_exec(cmd, hint, 0, ins, [out], s1) == 0     # first occurrence: on the spot
    L.stream_wait(s1)
    want = out.numpy().copy()
    x = [rng.standard_normal(1000).astype(np.float32) for _ in range(2)]
    for trip in range(3):
        out2 = on(1, np.full((n, h, w, k), -9, np.float32))
        t = [on(d, x[d]) for d in range(2)]
        assert L.cmd_exec(cmd, hint, 0, ins, [out2], s1) == 0   # recorded on s1 (look-ahead)
        L.dll.nnc_mi355x_debug_peephole_launch_delay_us(300000)
        loader = threading.Thread(target=lambda: on(0, np.zeros(16, np.float32)))   # a host-to-device copy: launches every recorded command
        loader.start()
        time.sleep(0.1)                                          # the loader sits inside the launch window of s1's convolution
        # device 0's collective is recorded first (its stream is not s1), then device 1's stream has to be resolved: that waits for the
        # loader's launch, whose own stream_of() wants the collectives' mutex as soon as one is pending
        assert L.cmd_exec(nnc.generic_cmd("COMM_ALLREDUCE_FORWARD"), nnc.NO_HINT, 0, t, t, s1) == 0
        loader.join()
        L.dll.nnc_mi355x_debug_peephole_launch_delay_us(0)
        L.stream_wait(s1); L.stream_wait(None)
        np.testing.assert_array_equal(out2.numpy(), want)
        for d in range(2):
            np.testing.assert_allclose(t[d].numpy(), x[0].astype(np.float64) + x[1], rtol=1e-6, atol=1e-6)
    L.stream_free(s1)
    print("OK")
""")


def test_comm_command_while_a_foreign_thread_launches_this_streams_recorded_command(emu_lib):
    # comm_exec resolved its streams with g_comm_mutex held -- stream_of() waits for a recorded command another thread is enqueueing
    # (peephole.cpp wait_launching) while that thread's launch waits for the mutex to flush the collective already recorded: a deadlock.
    env = dict(os.environ, NNC_EMU_DEVICE_COUNT="4")
    from conftest import emu_so
    so = emu_so()
    r = subprocess.run([sys.executable, "-c", LOCK_ORDER_SCRIPT % (ROOT, so)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


STRESS_SCRIPT = textwrap.dedent("""
    import ctypes as C, sys, threading, time, numpy as np
    sys.path.insert(0, %r)
    from ccv_amd import nnc
    L = nnc.load(%r)
    D = L.device_count()
    assert D == 8
    rng = np.random.default_rng(9)
    def on(dev, arr):
        return L.tensor(nnc.GPU_TENSOR_NHWC(dev, nnc.CCV_32F, *arr.shape), arr)
    streams = [L.stream_new(0), L.stream_new(1), L.stream_new(2)]      # three stream contexts, on three devices
    ar = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
    stop = [False]
    def loader():                                                       # a data-loader thread: copies and frees, each an order-observing point
        k = 0
        while not stop[0]:
            t = on(k %% D, np.zeros(64, np.float32)); t.free(); k += 1
    th = threading.Thread(target=loader); th.start()
    c0, g0 = C.c_long(), C.c_long(); L.dll.nnc_mi355x_comm_stats(C.byref(c0), C.byref(g0))
    for step in range(2):
        xs = [[rng.standard_normal(17 + 3 * m).astype(np.float32) for _ in range(D)] for m in range(42)]   # 42 gradient tensors x 8 devices = 336 records
        ts = [[on(d, xs[m][d]) for d in range(D)] for m in range(42)]
        for m in range(42):
            assert L.cmd_exec(ar, nnc.NO_HINT, 0, ts[m], ts[m], streams[m %% 3]) == 0
        for s in streams:
            L.stream_wait(s)
        for m in range(42):
            want = sum(x.astype(np.float64) for x in xs[m])
            for d in range(D):
                np.testing.assert_allclose(ts[m][d].numpy(), want, rtol=1e-6, atol=1e-6)
    stop[0] = True; th.join()
    c1, g1 = C.c_long(), C.c_long(); L.dll.nnc_mi355x_comm_stats(C.byref(c1), C.byref(g1))
    assert c1.value - c0.value == 2 * 336, c1.value - c0.value
    assert g1.value - g0.value >= 2 * 3, g1.value - g0.value          # a queue of 100: at least three forced group launches per step
    for s in streams:
        L.stream_free(s)
    print("OK", g1.value - g0.value)
""")


def test_comm_queue_overflow_on_eight_devices_three_streams_with_a_loader_thread(emu_lib):
    """>= 300 COMM records per step on 8 emulated devices from 3 stream contexts while a second thread copies and frees (every
    copy / free flushes the queue from that thread); the queue is cut to 100 records so that it overflows mid-step.  Every sum exact."""
    from conftest import emu_so
    env = dict(os.environ, NNC_EMU_DEVICE_COUNT="8", NNC_MI355X_COMM_MAX_PENDING="100")
    r = subprocess.run([sys.executable, "-c", STRESS_SCRIPT % (ROOT, emu_so())], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
