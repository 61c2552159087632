#!/usr/bin/env python
"""HBM rate of the element-wise / pooling rows on VGG-D-sized tensors (256 x 224 x 224 x 64 floats = 3.29 GB):
RELU forward (2|x| bytes), RELU backward (3|x|), EWSUM, 2x2/2 max pool forward / backward, with and without the grid cap of grid_for().
usage: tools/ew_bw_bench.py > gpurun_out/ew_bw_bench.txt"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccv_amd import nnc

L = nnc.load()
st = L.stream_new(0)
L.dll.nnc_mi355x_tune_set.argtypes = [C.c_char_p, C.c_long]
L.dll.nnc_mi355x_event_elapsed_ms.restype = C.c_float
L.dll.nnc_mi355x_event_new.restype = C.c_void_p
L.dll.nnc_mi355x_event_record.argtypes = [C.c_void_p, C.c_void_p]
L.dll.nnc_mi355x_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]


def tens(*dims, fill=False):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, dims, 0))
    if fill:
        rng = np.random.default_rng(0)
        block = (rng.random(1 << 22, dtype=np.float32) - 0.5)
        t.upload(np.resize(block, int(np.prod(dims))).reshape(dims))
    return t


def timed(label, nbytes, fn, reps=6):
    for _ in range(2):
        fn()
    L.stream_wait(st)
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
    L.dll.nnc_mi355x_event_record(e0, st)
    for _ in range(reps):
        fn()
    L.dll.nnc_mi355x_event_record(e1, st)
    ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
    print("%-52s %8.3f ms  %6.2f TB/s" % (label, ms, nbytes / ms / 1e9))
    sys.stdout.flush()


N, H, W, Cc = 256, 224, 224, 64
x, y, g, h = tens(N, H, W, Cc, fill=True), tens(N, H, W, Cc), tens(N, H, W, Cc, fill=True), tens(N, H, W, Cc)
nb = 4.0 * N * H * W * Cc
# the clocks settle over the first seconds of load: every setting is measured in several rounds, best round reported
def rate(nbytes, fn, reps=6):
    L.stream_wait(st)
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
    L.dll.nnc_mi355x_event_record(e0, st)
    for _ in range(reps):
        fn()
    L.dll.nnc_mi355x_event_record(e1, st)
    return nbytes / (L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps) / 1e9


# (The sweep that chose the current shape of ew_map_kernel -- grid-stride capped at 8 / 16 / 32 / 64 workgroups per CU, with and without
# non-temporal accesses, against a full grid -- is profiles/r02_v8_ew_bw_bench.txt; the kernel now always takes the full-grid form.
# GRID_WG_PER_CU still governs the grid-stride kernels behind grid_for(): the pooling kernels below.)
p, gp = tens(N, H // 2, W // 2, Cc), tens(N, H // 2, W // 2, Cc, fill=True)
hint = nnc.HINT((2, 2), (0, 0))
cases = [
    ("relu forward (2|x|)", 2 * nb, lambda: L.cmd_exec(nnc.CMD_RELU_FORWARD(), nnc.NO_HINT, 0, [x], [y], st)),
    ("relu backward (3|x|)", 3 * nb, lambda: L.cmd_exec(nnc.CMD_RELU_BACKWARD(), nnc.NO_HINT, 0, [g, None, y], [h], st)),
    ("ewsum of two (3|x|)", 3 * nb, lambda: L.cmd_exec(nnc.CMD_EWSUM_FORWARD(), nnc.NO_HINT, 0, [x, g], [h], st)),
    ("max pool 2x2/2 forward (1.25|x|)", 1.25 * nb, lambda: L.cmd_exec(nnc.CMD_MAX_POOL_FORWARD(2, 2), hint, 0, [x], [p], st)),
    ("max pool 2x2/2 backward (2.5|x|)", 2.5 * nb, lambda: L.cmd_exec(nnc.CMD_MAX_POOL_BACKWARD(2, 2), hint, 0, [gp, x, p], [h], st)),
]
best = {}
for rnd in range(4):
    for cap in (8, 0):
        L.dll.nnc_mi355x_tune_set(b"GRID_WG_PER_CU", cap)
        for label, nbytes, fn in cases:
            fn()
            best[(label, cap)] = max(best.get((label, cap), 0.0), rate(nbytes, fn))
for label, _, _ in cases:
    print("%-36s grid cap 8 per CU: %5.2f TB/s   no cap: %5.2f TB/s" % (label, best[(label, 8)], best[(label, 0)]))
