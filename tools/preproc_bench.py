#!/usr/bin/env python
"""Throughput of the batch pre-process kernels on the GPU box: 256 RGB 8-bit images 480x480 -> 224x224 (area), plus the
reference's ccv_resample on one host core for scale.  Prints images/s and effective GB/s (bytes read + written)."""
import ctypes as C
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ccv_amd import nnc  # noqa: E402
import test_preproc as T  # noqa: E402


def main():
    L = nnc.load()
    n, src, dst = 256, (480, 480, 3), (224, 224)
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, src, dtype=np.uint8) for _ in range(n)]
    L.dll.nnc_mi355x_resample_batch.argtypes = [C.c_void_p, T.ImageBatch, C.c_void_p, T.ImageBatch, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
    pa, da, _ = T.dev_batch(L, imgs)
    pb, db, _ = T.dev_batch(L, [np.zeros(dst + (3,), np.uint8) for _ in range(n)])
    st = L.stream_new(0)
    rs, cs = dst[0] / src[0], dst[1] / src[1]
    for kind, name in ((T.AREA, "area 8u"),):
        for _ in range(3):
            L.dll.nnc_mi355x_resample_batch(pa, da, pb, db, n, rs, cs, kind, st)
        L.stream_wait(st)
        e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
        reps = 20
        L.dll.nnc_mi355x_event_record(e0, st)
        for _ in range(reps):
            L.dll.nnc_mi355x_resample_batch(pa, da, pb, db, n, rs, cs, kind, st)
        L.dll.nnc_mi355x_event_record(e1, st)
        ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
        by = n * (src[0] * src[1] * 3 + dst[0] * dst[1] * 3)
        print("resample %s: %d x %s -> %s  %.3f ms  %.0f images/s  %.1f GB/s algorithmic" % (name, n, src, dst, ms, n / (ms * 1e-3), by / (ms * 1e-3) / 1e9))
    p = os.path.join(ROOT, "oracle", "_ref", "libccv_classic.so")
    if os.path.exists(p):
        R = C.CDLL(p)
        R.ccv_dense_matrix_new.restype = C.c_void_p
        R.ccv_dense_matrix_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        R.ccv_resample.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_double, C.c_int]
        R.ccv_matrix_free.argtypes = [C.c_void_p]
        t0 = time.time()
        k = 64
        for i in range(k):
            T.ref_resample(R, imgs[i], np.uint8, rs, cs, T.AREA)
        dt = time.time() - t0
        print("reference ccv_resample, 1 host core (incl. ctypes marshalling): %.0f images/s" % (k / dt))


if __name__ == "__main__":
    main()
