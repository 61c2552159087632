#!/usr/bin/env python
"""NCHW <-> NHWC re-layout (FORMAT_TRANSFORM) rates on the ResNet-50 / DawnNet activation shapes.
usage: tools/transpose_bench.py > gpurun_out/transpose_bench.txt"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccv_amd import nnc

L = nnc.load()
st = L.stream_new(0)
L.dll.nnc_mi355x_event_elapsed_ms.restype = C.c_float
L.dll.nnc_mi355x_event_new.restype = C.c_void_p
L.dll.nnc_mi355x_event_record.argtypes = [C.c_void_p, C.c_void_p]
L.dll.nnc_mi355x_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]


def rate(nbytes, fn, reps=10):
    for _ in range(3):
        fn()
    L.stream_wait(st)
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
    L.dll.nnc_mi355x_event_record(e0, st)
    for _ in range(reps):
        fn()
    L.dll.nnc_mi355x_event_record(e1, st)
    ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
    return ms, nbytes / ms / 1e9


cmd = nnc.generic_cmd("FORMAT_TRANSFORM_FORWARD")
for dt, es in ((nnc.CCV_32F, 4), (nnc.CCV_16F, 2)):
    for n, c, hw in ((256, 64, 56), (256, 128, 28), (256, 256, 14), (256, 512, 7), (256, 32, 112), (512, 128, 32), (512, 512, 4)):
        a = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, dt, (n, c, hw, hw), 0))
        b = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, dt, (n, hw, hw, c), 0))
        nb = 2.0 * es * n * c * hw * hw
        ms1, r1 = rate(nb, lambda: L.cmd_exec(cmd, nnc.NO_HINT, 0, [a], [b], st))
        ms2, r2 = rate(nb, lambda: L.cmd_exec(cmd, nnc.NO_HINT, 0, [b], [a], st))
        print("%s %4d x %4d x %3d^2  (%6.1f MB)  NCHW->NHWC %7.3f ms %5.2f TB/s   NHWC->NCHW %7.3f ms %5.2f TB/s" % ("f32" if es == 4 else "f16", n, c, hw, nb / 2e6, ms1, r1, ms2, r2))
        sys.stdout.flush()
        del a, b
