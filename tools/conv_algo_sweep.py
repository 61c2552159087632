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

    print("%-22s %9s %9s %7s | %9s %9s %7s" % ("layer (hw, C->K)", "fwd gemm", "fwd wino", "x", "bwd gemm", "bwd wino", "x"))
    for hw, c, k in LAYERS:
        a, w, b, bias = mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(batch, hw, hw, k), mk(k)
        g, h, dw, db = mk(batch, hw, hw, k), mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, bias, g], s)
        hint = nnc.HINT((1, 1), (1, 1))
        row = []
        for fwd in (True, False):
            t = []
            for algo in (0, 1):
                cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c) if fwd else nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
                cmd.algorithm = algo
                t.append(timed(cmd, hint, [a, w, bias] if fwd else [g, a, w], [b] if fwd else [h, dw, db]))
            row += [t[0], t[1], t[0] / t[1]]
        print("%-22s %9.3f %9.3f %7.2f | %9.3f %9.3f %7.2f" % ("%d, %d->%d" % (hw, c, k), *row), flush=True)
        for t in (a, w, b, bias, g, h, dw, db):
            t.free()


if __name__ == "__main__":
    main()
