#!/usr/bin/env python
"""Time every 3x3 VGG-D conv layer on the MI355X under both algorithms of the conv rows (0 = implicit GEMM,
1 = Winograd F(4x4,3x3)), forward and backward, HIP-event timed.  Feeds the wino_preferred() rule of cmd_conv.cpp.
usage: python tools/conv_algo_sweep.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccv_amd import nnc

LAYERS = [(223, 64, 64), (111, 64, 128), (111, 128, 128), (55, 128, 256), (55, 256, 256), (27, 256, 512), (27, 512, 512), (13, 512, 512)]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    L = nnc.load()
    s = L.stream_new(0)
    F = nnc.CCV_32F
    mk = lambda *d: L.tensor(nnc.GPU_TENSOR_NHWC(0, F, *d))
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()

    def timed(cmd, hint, ins, outs, reps=3):
        L.cmd_exec(cmd, hint, 0, ins, outs, s)
        L.dll.nnc_mi355x_event_record(e0, s)
        for _ in range(reps):
            L.cmd_exec(cmd, hint, 0, ins, outs, s)
        L.dll.nnc_mi355x_event_record(e1, s)
        return L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps

    algos = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
    print("batch %d; ms per command under cmd.algorithm = %s (0 implicit GEMM, 1 Winograd via HBM, 2 fused Winograd fwd / dgrad); [..] = MFMA TFLOP/s of the Winograd forms (direct FLOPs / 4)" % (batch, algos))
    for hw, c, k in LAYERS:
        a, w, b, bias = mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(batch, hw, hw, k), mk(k)
        g, h, dw, db = mk(batch, hw, hw, k), mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, bias, g], s)
        hint = nnc.HINT((1, 1), (1, 1))
        direct = 2.0 * batch * hw * hw * k * 9 * c
        for fwd in (True, False):
            out = []
            for algo in algos:
                cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c) if fwd else nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
                cmd.algorithm = algo
                ms = timed(cmd, hint, [a, w, bias] if fwd else [g, a, w], [b] if fwd else [h, dw, db])
                out.append("a%d %8.3f" % (algo, ms) + (" [%5.1f]" % ((1 if fwd else 2) * direct / 4 / (ms * 1e-3) / 1e12) if algo else ""))
            print("%-18s %-4s " % ("%d, %d->%d" % (hw, c, k), "fwd" if fwd else "bwd") + "   ".join(out), flush=True)
        for t in (a, w, b, bias, g, h, dw, db):
            t.free()


if __name__ == "__main__":
    main()
