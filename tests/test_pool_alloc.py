"""Device memory behind cumalloc / cufree (nnc_mi355x_malloc / _free): the device's memory pool (ccv_amd/csrc/device_rt.cpp; by default a free drains the
device like hipFree and returns the block to the pool, NNC_MI355X_POOL_ALLOC=2 only queues it).  What the reference's allocator layer above it expects (lib/nnc/ccv_nnc_xpu_alloc.c, lib/nnc/gpu/ccv_nnc_compat.cu cumalloc /
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
        assert pool_counts(lib)[0] - a0 >= 12
    finally:
        lib.stream_free(s1)
        lib.stream_free(s2)


@pytest.mark.gpu
def test_queued_free_mode_does_not_wait_for_the_device_and_the_pool_keeps_the_memory():
    """NNC_MI355X_POOL_ALLOC=2 (opt-in: the free is only queued): a free issued behind milliseconds of queued fills returns in well under a millisecond (the
    queue is still draining when it does), and the pool's reserved bytes do not shrink when memory is handed back.  Its own process: the mode is read once."""
    code = r'''
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ccv_amd import nnc
from harness import make_tensors
lib = nnc.load()
F = np.float32
def reserved():
    a, r, res, used = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    lib.dll.nnc_mi355x_debug_pool_counts(C.byref(a), C.byref(r), C.byref(res), C.byref(used))
    return res.value
(big,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(64 << 20, F)])
(t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(1 << 20, F)])
stream = lib.stream_new(0)
lib.stream_wait(stream)
for _ in range(400):
    assert lib.cmd_exec(nnc.CMD_SET_FORWARD(2.0), nnc.NO_HINT, 0, [], [big], stream) == 0
r0 = reserved()
t1 = time.perf_counter(); t.free(); t2 = time.perf_counter()
lib.stream_wait(stream)
t3 = time.perf_counter()
assert t3 - t2 > 10 * (t2 - t1), "the queue had drained before the free returned: free %%g s, rest of the queue %%g s" %% (t2 - t1, t3 - t2)
assert t2 - t1 < 2e-3, "free took %%g s" %% (t2 - t1)
assert reserved() >= r0 > 0
print("ok")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, NNC_MI355X_POOL_ALLOC="2"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_pressure_callbacks_run_before_an_allocation_is_given_up():
    """CPU tier (emulator): the first pool allocation fails (EMU_POOL_FAIL_NEXT=1); the callback registered through nnc_mi355x_register_mem_pressure -- the
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
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, EMU_POOL_FAIL_NEXT="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
