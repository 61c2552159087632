#!/usr/bin/env python
"""cumalloc / cufree (nnc_mi355x_malloc / _free) with the stream-ordered pool against the blocking hipMalloc / hipFree pair (NNC_MI355X_POOL_ALLOC=0): host
time per allocate + free of a few sizes on an idle device and behind a busy stream -- what a dynamic graph's tensor churn costs when it misses the host's own
free lists (lib/nnc/ccv_nnc_xpu_alloc.c).  One process per setting.  usage: python tools/alloc_bench.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one():
    import numpy as np
    from ccv_amd import nnc
    from harness import make_tensors
    L = nnc.load()
    (big,) = make_tensors(L, nnc.GPU_MEMORY, [np.zeros(64 << 20, np.float32)])
    stream = L.stream_new(0)
    mode = "pool" if os.environ.get("NNC_MI355X_POOL_ALLOC", "1") != "0" else "hipMalloc/hipFree"
    for busy in (0, 1):
        for mb in (1, 64, 1024):
            n = mb << 20
            for _ in range(3):
                L.free(0, L.malloc(0, n))  # warm
            L.stream_wait(stream)
            reps = 50
            t_alloc = t_free = 0.0
            for _ in range(reps):
                if busy:
                    for _ in range(20):  # ~1 ms of queued fills in front of every pair
                        L.cmd_exec(nnc.CMD_SET_FORWARD(1.0), nnc.NO_HINT, 0, [], [big], stream)
                t0 = time.perf_counter()
                p = L.malloc(0, n)
                t1 = time.perf_counter()
                L.free(0, p)
                t2 = time.perf_counter()
                t_alloc += t1 - t0
                t_free += t2 - t1
            L.stream_wait(stream)
            print("%-18s %s  %5d MB   allocate %8.1f us   free %8.1f us" % (mode, "behind a busy stream" if busy else "idle device         ", mb, 1e6 * t_alloc / reps, 1e6 * t_free / reps))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for v in ("0", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, NNC_MI355X_POOL_ALLOC=v))
