#!/usr/bin/env python
"""Time the Winograd-via-HBM conv commands of the low-channel VGG-D layers on the MI355X for a sweep of WINO_SLICE_KB (images
per slice sized so that the transformed images V + M of one slice stay in the 256 MB Infinity Cache), forward and backward,
HIP-event timed on the stream.  Feeds the default of TUNE_WINO_SLICE_KB (ccv_amd/csrc/device_rt.cpp).
usage: python tools/wino_slice_sweep.py [batch] [algo]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccv_amd import nnc

LAYERS = [(223, 64, 64), (111, 64, 128), (111, 128, 128), (55, 128, 256), (55, 256, 256), (27, 256, 512), (27, 512, 512), (13, 512, 512)]
SLICES_MB = [0, 24, 48, 64, 96, 128, 160, 192, 256]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    algo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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

    print("batch %d algorithm %d; columns: WINO_SLICE_KB in MB (0 = whole batch); ms per command" % (batch, algo))
    print("%-18s %-4s " % ("layer", "") + " ".join("%8d" % m for m in SLICES_MB))
    for hw, c, k in LAYERS:
        a, w, b, bias = mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(batch, hw, hw, k), mk(k)
        g, h, dw, db = mk(batch, hw, hw, k), mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, bias, g], s)
        hint = nnc.HINT((1, 1), (1, 1))
        for fwd in (True, False):
            row = []
            for mb in SLICES_MB:
                L.tune_set("WINO_SLICE_KB", mb * 1024)
                cmd = nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c) if fwd else nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
                cmd.algorithm = algo
                row.append(timed(cmd, hint, [a, w, bias] if fwd else [g, a, w], [b] if fwd else [h, dw, db]))
            print("%-18s %-4s " % ("%d, %d->%d" % (hw, c, k), "fwd" if fwd else "bwd") + " ".join("%8.3f" % t for t in row), flush=True)
        L.tune_set("WINO_SLICE_KB", 0)
        for t in (a, w, b, bias, g, h, dw, db):
            t.free()


if __name__ == "__main__":
    main()
