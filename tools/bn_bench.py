#!/usr/bin/env python
"""Batch norm (training) forward / backward on the [256][C][H][W] tensors of ResNet-50 v1d at batch 256, fp32 and CCV_16F:
the cluster kernels (x read once: a cluster of workgroups per channel keeps the channel in registers, cmd_norm.cpp) against the plane kernels
(BN_CLUSTER=0: statistics pass + apply pass).  Rates are ALGORITHMIC bytes (forward 2 |x|, backward 3 |x|) / time.
usage: tools/bn_bench.py > gpurun_out/bn_bench.txt"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccv_amd import nnc

L = nnc.load(os.environ.get("NNC_BENCH_LIB"))  # (NNC_BENCH_LIB: a dry run of this script on the CPU emulator build)
st = L.stream_new(0)
d = L.dll
d.nnc_mi355x_event_elapsed_ms.restype = C.c_float
d.nnc_mi355x_event_new.restype = C.c_void_p
d.nnc_mi355x_event_record.argtypes = [C.c_void_p, C.c_void_p]
d.nnc_mi355x_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
N = int(os.environ.get("BN_BENCH_N", "256"))
SHAPES = [(32, 112, 112), (64, 112, 112), (64, 56, 56), (256, 56, 56), (128, 56, 56), (128, 28, 28), (512, 28, 28), (256, 28, 28), (256, 14, 14), (1024, 14, 14), (512, 14, 14), (512, 7, 7), (2048, 7, 7)]
if os.environ.get("BN_BENCH_TINY"):
    SHAPES = [(3, 8, 8), (2, 7, 7)]


def tens(dt, *dims, fill=False):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, dt, dims, 0))
    if fill:  # on the device (RANDOM_NORMAL_FORWARD; half tensors through the staged fp32 image): no gigabytes through numpy and PCIe per shape
        assert L.cmd_exec(nnc._blas_a("RANDOM_NORMAL_FORWARD", 1.0, 0.25), nnc.NO_HINT, 0, [], [t], st) == 0
    return t


def rate(fn, reps=5):
    for _ in range(2):
        fn()
    L.stream_wait(st)
    best = 1e9
    for _ in range(3):
        e0, e1 = d.nnc_mi355x_event_new(), d.nnc_mi355x_event_new()
        d.nnc_mi355x_event_record(e0, st)
        for _ in range(reps):
            fn()
        d.nnc_mi355x_event_record(e1, st)
        best = min(best, d.nnc_mi355x_event_elapsed_ms(e0, e1) / reps)
    return best


tot = {}
print("%-22s %-4s | %-31s | %-31s" % ("[256][C][H][W]", "", "forward ms (TB/s of 2|x|): cluster / planes", "backward ms (TB/s of 3|x|): cluster / planes"))
for dt, name, esz in ((nnc.CCV_32F, "f32", 4), (nnc.CCV_16F, "f16", 2)):
    for (c, h, w) in SHAPES:
        x, g = tens(dt, N, c, h, w, fill=True), tens(dt, N, c, h, w, fill=True)
        y, hh = tens(dt, N, c, h, w), tens(dt, N, c, h, w)
        s = [tens(nnc.CCV_32F, c, fill=True) for _ in range(4)]
        L.stream_wait(st)
        s[3].upload(np.abs(s[3].numpy()) + 0.5)
        sm, si, ds, db = [tens(nnc.CCV_32F, c) for _ in range(4)]
        fw = nnc.CMD_BATCH_NORM_FORWARD(1e-4, 0, 0.9, 0, 2, 3)
        bw = nnc.CMD_BATCH_NORM_BACKWARD(1e-4, 0, 0.9, 0, 2, 3)
        f = lambda: L.cmd_exec(fw, nnc.NO_HINT, 0, [x] + s, [y, s[2], s[3], sm, si], st)
        b = lambda: L.cmd_exec(bw, nnc.NO_HINT, 0, [g] + [None] * 4 + [x, s[0]] + [None] * 6 + [sm, si], [hh, ds, db], st)
        nb = float(N) * c * h * w * esz
        row = []
        for mode in (1, 0):
            L.tune_set("BN_CLUSTER", mode)
            n0 = d.nnc_mi355x_debug_bn_cluster_launches()
            tf, tb = rate(f), rate(b)
            took = d.nnc_mi355x_debug_bn_cluster_launches() > n0
            row.append((tf, tb, took))
            k = (name, mode)
            tot[k] = (tot.get(k, (0, 0, 0, 0))[0] + tf, tot.get(k, (0, 0, 0, 0))[1] + tb, tot.get(k, (0, 0, 0, 0))[2] + 2 * nb, tot.get(k, (0, 0, 0, 0))[3] + 3 * nb)
        L.tune_set("BN_CLUSTER", 1)
        (cf, cb, took), (pf, pb, _) = row
        print("%-22s %-4s | %7.3f (%5.2f) / %7.3f (%5.2f) %s | %7.3f (%5.2f) / %7.3f (%5.2f)" % ("%d x %d x %d" % (c, h, w), name, cf, 2 * nb / cf / 1e9, pf, 2 * nb / pf / 1e9, " " if took else "*", cb, 3 * nb / cb / 1e9, pb, 3 * nb / pb / 1e9))
        sys.stdout.flush()
        for t in [x, g, y, hh] + s + [sm, si, ds, db]:
            t.free()
print("(* = the cluster kernels did not take this shape: planes that are not whole 16-byte chunks)")
for (name, mode), (tf, tb, bf, bb) in sorted(tot.items()):
    print("sum over the shapes, %s, %s: forward %.3f ms = %.2f TB/s of algorithmic bytes, backward %.3f ms = %.2f TB/s" % (name, "cluster" if mode else "planes ", tf, bf / tf / 1e9, tb, bb / tb / 1e9))
