"""The pinned staging ring (include/nnc_mi355x.h: nnc_mi355x_staging_ring_*) -- the host side of the GPU data pipeline: batches
assembled in pinned slots, copied asynchronously on the ring's own stream, consumed by kernels on another stream, slots reused in a cycle.  Every batch
that comes out of the consumer must be the batch that went in, whatever the interleaving; bad slots are refused."""
import ctypes as C
import numpy as np
from ccv_amd import nnc


def _bind(L):
    d = L.dll
    d.nnc_mi355x_staging_ring_new.restype = C.c_void_p
    d.nnc_mi355x_staging_ring_new.argtypes = [C.c_int, C.c_int, C.c_size_t]
    for f in (d.nnc_mi355x_staging_ring_host, d.nnc_mi355x_staging_ring_device):
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_int]
    d.nnc_mi355x_staging_ring_submit.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    d.nnc_mi355x_staging_ring_acquire.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    d.nnc_mi355x_staging_ring_release.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    d.nnc_mi355x_staging_ring_free.argtypes = [C.c_void_p]
    return d


def test_batches_survive_the_ring(backend):
    L = backend
    d = _bind(L)
    n, slots, batches = 4096, 3, 11
    ring = d.nnc_mi355x_staging_ring_new(0, slots, n * 4)
    assert ring
    stream = L.stream_new(0)
    rng = np.random.default_rng(2)
    data = [rng.standard_normal(n).astype(np.float32) for _ in range(batches)]
    outs = [L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, n)) for _ in range(batches)]
    scale = nnc.CMD_SCALAR_MUL_FORWARD(2.0)
    t = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, n))
    own = t.struct.data
    try:
        # the loader runs `slots` - 1 batches ahead of the consumer
        def submit(b):
            s = b % slots
            host = d.nnc_mi355x_staging_ring_host(ring, s)          # blocks until the previous copy out of this pinned buffer is done
            C.memmove(host, data[b].ctypes.data, n * 4)
            assert d.nnc_mi355x_staging_ring_submit(ring, s, n * 4) == 1
        for b in range(min(slots - 1, batches)):
            submit(b)
        for b in range(batches):
            if b + slots - 1 < batches:
                submit(b + slots - 1)
            s = b % slots
            assert d.nnc_mi355x_staging_ring_acquire(ring, s, stream) == 1
            # consume: out[b] = 2 * (the slot's device buffer), through a tensor struct pointed at the ring's memory
            t.struct.data = d.nnc_mi355x_staging_ring_device(ring, s)
            assert L.cmd_exec(scale, nnc.NO_HINT, 0, [t], [outs[b]], stream) == 0
            assert d.nnc_mi355x_staging_ring_release(ring, s, stream) == 1
        L.stream_wait(stream)
        for b in range(batches):
            np.testing.assert_array_equal(outs[b].numpy(), data[b] * np.float32(2))
        assert d.nnc_mi355x_staging_ring_submit(ring, slots, 16) == 0 and d.nnc_mi355x_staging_ring_submit(ring, 0, n * 4 + 1) == 0
        assert not d.nnc_mi355x_staging_ring_host(ring, -1) and not d.nnc_mi355x_staging_ring_device(ring, slots)
    finally:
        t.struct.data = own
        L.stream_free(stream)
        d.nnc_mi355x_staging_ring_free(ring)
