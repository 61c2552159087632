#!/usr/bin/env python
"""nnc_mi355x_memcpy (= cumemcpy) on PAGEABLE host memory, both directions: what ccv_nnc_tensor_write / _read (lib/nnc/ccv_nnc_tensor_io.c:28-133) and
DATA_TRANSFER of CPU tensors cost per GB.  Round 5 ran it with a hand-made pinned ring behind NNC_MI355X_STAGED_COPY (profiles/r05_v1_memcpy_pageable.txt:
slower than the runtime's own blocking copy, removed); the tool now times the blocking copy alone.  usage: python tools/memcpy_bench.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    from ccv_amd import nnc
    L = nnc.load()
    for mb in (16, 256, 2048):
        n = mb << 20
        host = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)  # pageable, touched
        back = np.zeros(n, dtype=np.uint8)
        dev = L.malloc(0, n)
        for rep in range(2):
            t0 = time.perf_counter()
            L.memcpy(dev, nnc.GPU_MEMORY, host.ctypes.data, nnc.CPU_MEMORY, n)
            t1 = time.perf_counter()
            L.memcpy(back.ctypes.data, nnc.CPU_MEMORY, dev, nnc.GPU_MEMORY, n)
            t2 = time.perf_counter()
        ok = bool((back == host).all())
        print("blocking copy %s  %5d MB   host->device %6.2f GB/s   device->host %6.2f GB/s   round trip %s" % ("", mb, n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9, "bit-exact" if ok else "DIFFERS"))
        L.free(0, dev)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        subprocess.run([sys.executable, os.path.abspath(__file__), "one"])
