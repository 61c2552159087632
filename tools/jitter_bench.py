#!/usr/bin/env python
"""Throughput of the data pipeline's pixel stage on the GPU box: 256 RGB 8-bit images of ImageNet-like
sizes -> the trainer's 224 x 224 x 3 batch tensor (random resize 256..480 on the short side, random crop, mirror, normalise), the
decisions drawn on the host, one jitter kernel per batch; H2D of the raw images from pinned memory included in the second
figure.  Next to it: the reference's own pixel functions on one host core.  Prints images/s."""
import ctypes as C
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ccv_amd import nnc  # noqa: E402
import test_jitter as J  # noqa: E402


def main():
    sys.stdout.reconfigure(line_buffering=True)
    L = nnc.load()
    n, size = 256, (224, 224)
    rng = np.random.default_rng(0)
    shapes = [(int(rng.integers(333, 500)), int(rng.integers(375, 640))) for _ in range(n)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]
    plans = [J.plan(rng, s[0], s[1], 256, 480, size, aspect=0.5) for s in shapes]
    descs = (J.JitterImage * n)()
    blobs, off = [], 0
    for i, (a, p) in enumerate(zip(imgs, plans)):
        rows, cols = a.shape[:2]
        step = (cols * 3 + 3) & ~3
        buf = np.zeros((rows, step), np.uint8)
        buf[:, :cols * 3] = a.reshape(rows, -1)
        x, y, sr, sc = p["slice"]
        descs[i] = J.JitterImage(off, rows, cols, step, x, y, sr, sc, p["resize"][0], p["resize"][1], p["crop"][0], p["crop"][1], int(p["flip"]))
        pad = (-buf.size) % 16
        blobs.append(buf.reshape(-1))
        blobs.append(np.zeros(pad, np.uint8))
        off += buf.size + pad
    host = np.concatenate(blobs)
    L.dll.nnc_mi355x_host_alloc.restype = C.c_void_p
    L.dll.nnc_mi355x_host_alloc.argtypes = [C.c_size_t]
    pinned = L.dll.nnc_mi355x_host_alloc(host.nbytes)  # pinned staging buffer (cuhostalloc)
    C.memmove(pinned, host.ctypes.data, host.nbytes)
    src = L.malloc(0, (host.nbytes + 127) & ~127)
    L.memcpy(src, nnc.GPU_MEMORY, host.ctypes.data, nnc.CPU_MEMORY, host.nbytes)
    out = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, nnc.CCV_32F, (n, 3) + size, 0))
    mean, std = (123.68, 116.779, 103.939), (58.393, 57.12, 57.375)
    params = J.JitterParams(size[0], size[1], 3, (C.c_float * 3)(*mean), (C.c_float * 3)(*[1.0 / s for s in std]), nnc.NCHW, nnc.CCV_32F)
    L.dll.nnc_mi355x_jitter_batch.argtypes = [C.c_void_p, C.POINTER(J.JitterImage), C.c_int, J.JitterParams, C.c_void_p, C.c_void_p]
    st = L.stream_new(0)
    for with_h2d in (False, True):
        def once():
            if with_h2d:
                L.memcpy(src, nnc.GPU_MEMORY, pinned, nnc.CPU_MEMORY, host.nbytes)  # blocking copy from pinned memory
            assert L.dll.nnc_mi355x_jitter_batch(src, descs, n, params, out.ptr, st) == 0
        for _ in range(3):
            once()
        L.stream_wait(st)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        L.stream_wait(st)
        dt = (time.perf_counter() - t0) / reps
        print("jitter batch (%s): %d images (%.0f MB raw) -> %dx%dx3 fp32 NCHW: %.3f ms  %.0f images/s (host table building included)" % (
            "raw images copied H2D every batch" if with_h2d else "raw images resident in HBM", n, host.nbytes / 1e6, size[0], size[1], dt * 1e3, n / dt))
    # the pinned staging ring (nnc_mi355x_staging_ring_*): the raw images of batch b + 1 are copied while batch b's kernel runs
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
    slots = 3
    ring = d.nnc_mi355x_staging_ring_new(0, slots, (host.nbytes + 127) & ~127)
    for fill in (False, True):
        # fill: the loader also assembles the batch in the pinned slot every time (a 160 MB memcpy on ONE host thread -- the dataframe's loader threads
        # would do this in parallel); not fill: the slots are assembled once, what is timed is copy + kernel overlapped
        for sl in range(slots):
            C.memmove(d.nnc_mi355x_staging_ring_host(ring, sl), host.ctypes.data, host.nbytes)
        reps = 12

        def submit(b):
            sl = b % slots
            hp = d.nnc_mi355x_staging_ring_host(ring, sl)
            if fill:
                C.memmove(hp, host.ctypes.data, host.nbytes)
            assert d.nnc_mi355x_staging_ring_submit(ring, sl, host.nbytes) == 1
        for phase in ("warm", "timed"):
            for b in range(slots - 1):
                submit(b)
            L.stream_wait(st)
            t0 = time.perf_counter()
            for b in range(reps):
                if b + slots - 1 < reps:
                    submit(b + slots - 1)
                sl = b % slots
                d.nnc_mi355x_staging_ring_acquire(ring, sl, st)
                assert d.nnc_mi355x_jitter_batch(d.nnc_mi355x_staging_ring_device(ring, sl), descs, n, params, out.ptr, st) == 0
                d.nnc_mi355x_staging_ring_release(ring, sl, st)
            L.stream_wait(st)
            dt = (time.perf_counter() - t0) / reps
        print("jitter batch through the pinned staging ring (%d slots, async H2D on its own stream%s): %.3f ms  %.0f images/s  (%.1f GB/s of raw images)" % (
            slots, ", slot assembled on one host thread every batch" if fill else "", dt * 1e3, n / dt, host.nbytes / dt / 1e9))
    d.nnc_mi355x_staging_ring_free(ring)
    p = os.path.join(ROOT, "oracle", "_ref", "libccv_classic.so")
    if os.path.exists(p):
        R = C.CDLL(p)
        R.ccv_dense_matrix_new.restype = C.c_void_p
        R.ccv_dense_matrix_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        R.ccv_resample.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_double, C.c_int]
        R.ccv_slice.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        R.ccv_flip.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
        R.ccv_shift.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        k = 32
        t0 = time.time()
        for i in range(k):
            J.ref_pipeline(R, imgs[i], plans[i], size, mean, [1.0 / s for s in std])
        print("reference pixel functions (ccv_slice / ccv_resample / ccv_flip + normalise), 1 host core incl. ctypes marshalling: %.0f images/s" % (k / (time.time() - t0)))


if __name__ == "__main__":
    main()
