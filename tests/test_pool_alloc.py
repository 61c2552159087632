"""Device memory behind cumalloc / cufree (nnc_mi355x_malloc / _free): a caching layer over hipMalloc (ccv_amd/csrc/device_rt.cpp: a free records an event
behind every stream that still has work in flight and keeps the block -- stream-ordered, no device drain (round 6) --; an allocation of the same rounded size
reuses a block whose events have completed).  What the reference's allocator layer above it expects (lib/nnc/ccv_nnc_xpu_alloc.c, lib/nnc/gpu/ccv_nnc_compat.cu cumalloc /
cufree / curegmp): memory that is safe to use by work queued after the allocation returned, a free that may be issued right behind queueing the last use, and
the registered pressure callbacks run before an allocation is given up."""
import ctypes as C
import os
import subprocess
import sys
import time
import numpy as np
import pytest
from ccv_amd import nnc
from harness import make_tensors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def pool_counts(lib):
    a, r, res, used = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    lib.dll.nnc_mi355x_debug_pool_counts(C.byref(a), C.byref(r), C.byref(res), C.byref(used))
    return a.value, r.value, res.value, used.value


def test_freed_memory_is_reused_in_stream_order_across_streams(backend):
    """A tensor is filled on stream 1 and freed right behind the command; the next allocation of the size (served from the pool: as a rule the same block)
    is filled on stream 2 with another value and read back.  Every element holds the second value -- the first stream's late kernel cannot land on top of it."""
    lib = backend
    n = 8 << 20
    s1, s2 = lib.stream_new(0), lib.stream_new(0)
    a0 = pool_counts(lib)[0]
    try:
        for trip in range(6):
            (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])
            for _ in range(4):  # a queue of work on stream 1 still running when the free is issued
                assert lib.cmd_exec(nnc.CMD_SET_FORWARD(1.0 + trip), nnc.NO_HINT, 0, [], [t], s1) == 0
            ptr = t.ptr
            t.free()
            (u,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full(n, -1, F)])
            assert lib.cmd_exec(nnc.CMD_SET_FORWARD(-5.0 - trip), nnc.NO_HINT, 0, [], [u], s2) == 0
            got = u.numpy()
            assert (got == F(-5.0 - trip)).all(), (trip, ptr == u.ptr)
            u.free()
        assert pool_counts(lib)[0] - a0 >= 10  # (all but the first allocation of the size come out of kept blocks)
    finally:
        lib.stream_free(s1)
        lib.stream_free(s2)


@pytest.mark.gpu
def test_a_freed_gigabyte_comes_back_without_the_driver(gpu_lib):
    """hipMalloc of 1 GB costs tens of milliseconds on the MI355X box (profiles/r05_v2_alloc_bench.txt); a block that has been freed once is handed out again
    in microseconds, and the bytes the layer holds do not grow when the same size goes round."""
    lib = gpu_lib
    n = 1 << 30
    p = lib.malloc(0, n)
    assert p
    lib.free(0, p)
    held0 = pool_counts(lib)[2]
    a0 = pool_counts(lib)[0]
    t0 = time.perf_counter()
    for _ in range(10):
        q = lib.malloc(0, n)
        assert q == p
        lib.free(0, q)
    dt = (time.perf_counter() - t0) / 10
    assert dt < 2e-3, "allocate + free of a kept 1 GB block took %g s" % dt
    assert pool_counts(lib)[0] - a0 == 10 and pool_counts(lib)[2] == held0


def test_pressure_callbacks_run_before_an_allocation_is_given_up():
    """CPU tier (emulator): the first device allocation fails (EMU_MALLOC_FAIL_NEXT=1); the callback registered through nnc_mi355x_register_mem_pressure -- the
    reference host registers ccv_nnc_xpu_alloc's drain and the stream contexts' workspace drains there (curegmp) -- runs once and the retry succeeds."""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ccv_amd import nnc
lib = nnc.load(os.path.join(%r, "tests", "emu", "_build", "libnnc_mi355x_emu.so"))
calls = []
CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)
cb = CB(lambda device, ctx: calls.append(device))
lib.dll.nnc_mi355x_register_mem_pressure.argtypes = [C.c_int, CB, C.c_void_p]
slot = lib.dll.nnc_mi355x_register_mem_pressure(0, cb, None)
p = lib.malloc(0, 1 << 20)
a, r = C.c_long(), C.c_long()
lib.dll.nnc_mi355x_debug_pool_counts(C.byref(a), C.byref(r), None, None)
assert p and calls == [0] and r.value == 1, (p, calls, r.value)
lib.free(0, p)
q = lib.malloc(0, 1 << 20)
assert q and calls == [0]
lib.free(0, q)
print("ok")
''' % (ROOT, ROOT, ROOT)
    so = os.path.join(ROOT, "tests", "emu", "_build", "libnnc_mi355x_emu.so")
    if not os.path.exists(so):
        pytest.skip("emulator library not built")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, EMU_MALLOC_FAIL_NEXT="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_the_kept_bytes_are_bounded():
    """CPU tier (emulator): a caller whose allocation sizes never repeat cannot grow the caching layer without bound -- beyond the cap (NNC_MI355X_POOL_KEEP_MB;
    half the device's memory by default) the oldest kept blocks go back to the driver.  Own process: the cap is read once."""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ccv_amd import nnc
lib = nnc.load(os.path.join(%r, "tests", "emu", "_build", "libnnc_mi355x_emu.so"))
def counts():
    a, r, res, used = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    lib.dll.nnc_mi355x_debug_pool_counts(C.byref(a), C.byref(r), C.byref(res), C.byref(used))
    return a.value, r.value, res.value, used.value
lib.dll.nnc_mi355x_debug_pool_trimmed.restype = C.c_long
for i in range(64):  # sizes that never repeat: without the bound every freed block would stay on the layer's lists for good
    p = lib.malloc(0, (1 << 20) * (3 + i))
    assert p
    lib.free(0, p)
    held = counts()[2]
    assert held <= (8 << 20) + (2 << 20) * 40, (i, held)  # the cap (8 MB) + at most the block just freed
assert counts()[3] == 0
assert lib.dll.nnc_mi355x_debug_pool_trimmed() >= 55, lib.dll.nnc_mi355x_debug_pool_trimmed()
a0 = counts()[0]
for i in range(4):  # a size that repeats within the cap is still served from its kept block
    p = lib.malloc(0, 4 << 20); assert p; lib.free(0, p)
assert counts()[0] - a0 >= 3
print("ok")
''' % (ROOT, ROOT, ROOT)
    so = os.path.join(ROOT, "tests", "emu", "_build", "libnnc_mi355x_emu.so")
    if not os.path.exists(so):
        pytest.skip("emulator library not built")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, NNC_MI355X_POOL_KEEP_MB="8"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_a_blocking_upload_into_reused_memory_is_not_overwritten_by_another_streams_late_kernels(backend):
    """The case that sank round 5's first allocator (a free merely QUEUED on the legacy stream): a block still being written by a queue of kernels on stream 1
    is freed; the next allocation of the size is filled by a BLOCKING host-to-device copy -- no stream of its own -- and read back at once.  The read-back holds
    the uploaded values: the allocation either got another block or waited for stream 1's events.  Then the same through a second stream and the legacy stream
    together (three streams in all)."""
    lib = backend
    n = 4 << 20
    s1, s2 = lib.stream_new(0), lib.stream_new(0)
    try:
        for trip in range(4):
            (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])
            for _ in range(6):
                assert lib.cmd_exec(nnc.CMD_SET_FORWARD(3.0 + trip), nnc.NO_HINT, 0, [], [t], s1) == 0
            t.free()                                                             # stream 1 is still writing t
            want = np.arange(n, dtype=F) + trip
            (u,) = make_tensors(lib, nnc.GPU_MEMORY, [want])                     # blocking upload into (as a rule) the block just freed
            assert np.array_equal(u.numpy(), want), trip
            for _ in range(3):
                assert lib.cmd_exec(nnc.CMD_SET_FORWARD(-1.0), nnc.NO_HINT, 0, [], [u], s2) == 0
            u.free()                                                             # stream 2 is still writing u
            (v,) = make_tensors(lib, nnc.GPU_MEMORY, [want + 1])
            assert lib.cmd_exec(nnc.CMD_SET_FORWARD(7.0 + trip), nnc.NO_HINT, 0, [], [v], None) == 0  # the legacy stream
            assert (v.numpy() == F(7.0 + trip)).all(), trip
            v.free()
    finally:
        lib.stream_free(s1)
        lib.stream_free(s2)


def test_freeing_twice_aborts():
    """CPU tier (emulator): a pointer that is already on the kept list is refused loudly (abort), where handing it to the driver would have left a dangling
    block on the list for the next allocation of its size."""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ccv_amd import nnc
lib = nnc.load(os.path.join(%r, "tests", "emu", "_build", "libnnc_mi355x_emu.so"))
p = lib.malloc(0, 1 << 20)
lib.free(0, p)
print("freed once", flush=True)
lib.free(0, p)
print("survived")
''' % (ROOT, ROOT, ROOT)
    so = os.path.join(ROOT, "tests", "emu", "_build", "libnnc_mi355x_emu.so")
    if not os.path.exists(so):
        pytest.skip("emulator library not built")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "freed once" in r.stdout and "survived" not in r.stdout and "double free" in r.stderr, r.stdout + r.stderr


def test_pool_trim_returns_the_kept_blocks(backend):
    """nnc_mi355x_pool_trim: every kept block goes back to the driver (the library calls it before RCCL communicators are created); blocks in use are untouched."""
    lib = backend
    lib.dll.nnc_mi355x_pool_trim.argtypes = [C.c_int]
    (keep,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full(1 << 18, 5, F)])
    for sz in (1 << 20, 3 << 20, 5 << 20):
        p = lib.malloc(0, sz)
        assert p
        lib.free(0, p)
    _, _, held, used = pool_counts(lib)
    assert held - used >= 9 << 20
    lib.dll.nnc_mi355x_pool_trim(-1)
    _, _, held, used = pool_counts(lib)
    assert held == used
    assert (keep.numpy() == 5).all()
    keep.free()


@pytest.mark.gpu
def test_a_free_behind_a_busy_stream_returns_without_draining_the_device(gpu_lib):
    """cufree behind a busy stream used to pay hipDeviceSynchronize (640 us and more: the queue's remaining run time).  Now it records one
    event per busy stream: the free returns while the stream's queue is still running (the stream wait behind it takes far longer than the free), and an
    allocation of the size right behind it is served without waiting -- from another block -- or, with only that block kept, after its events."""
    lib = gpu_lib
    ev, wt = C.c_long(), C.c_long()
    n = 256 << 20                                                                # 1 GB of floats: ~0.4 ms per fill
    s1 = lib.stream_new(0)
    try:
        (small,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(1 << 20, F)])
        (fill,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])
        lib.stream_wait(None)
        frees = []
        for trip in range(5):
            for _ in range(10):                                                  # ~4 ms of queued work on stream 1
                assert lib.cmd_exec(nnc.CMD_SET_FORWARD(1.0), nnc.NO_HINT, 0, [], [fill], s1) == 0
            assert lib.cmd_exec(nnc.CMD_SET_FORWARD(2.0), nnc.NO_HINT, 0, [], [small], s1) == 0
            lib.dll.nnc_mi355x_debug_pool_fences(C.byref(ev), C.byref(wt))
            e0 = ev.value
            t0 = time.perf_counter()
            small.free()
            t1 = time.perf_counter()
            lib.stream_wait(s1)
            t2 = time.perf_counter()
            lib.dll.nnc_mi355x_debug_pool_fences(C.byref(ev), C.byref(wt))
            assert ev.value > e0                                                 # the free found stream 1 busy and put an event behind it
            frees.append((t1 - t0, t2 - t1))
            (small,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(1 << 20, F)])
        print("free behind a busy stream: %s us; the stream's remaining queue: %s us" % ([round(a * 1e6, 1) for a, _ in frees], [round(b * 1e6) for _, b in frees]))
        best = min(a for a, _ in frees)
        assert best < 100e-6, frees                                               # (the drain it replaces: the queue's remaining milliseconds)
        assert max(b for _, b in frees) > 1e-3, frees                             # ... and the queue was indeed still running when the free returned
        fill.free(); small.free()
    finally:
        lib.stream_free(s1)
