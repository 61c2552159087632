#!/usr/bin/env python
"""Time the FILTER GRADIENT of the 3x3 VGG-D / ResNet / DawnNet layers on the MI355X: Winograd via HBM (cmd.algorithm = 1: wino_input + wino_outgrad +
36 GEMMs + final) against the fused form (algorithm = 2, wino_wgrad_fused.h), data gradient off (outputs[0] = NULL), HIP-event timed.  Feeds
TUNE_WINO_WGRAD_FUSED_MAX.   usage: python tools/wgrad_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccv_amd import nnc

LAYERS = [(223, 64, 64), (111, 64, 128), (111, 128, 128), (55, 128, 256), (55, 256, 256), (27, 256, 512), (56, 64, 64), (28, 128, 128), (32, 64, 128)]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    L = nnc.load()
    L.tune_set("WINO_WGRAD_FUSED_MAX", 1 << 20)  # measure the fused form on every eligible shape (the library's rule takes it up to 128 channels)
    s = L.stream_new(0)
    F = nnc.CCV_32F
    mk = lambda *d: L.tensor(nnc.GPU_TENSOR_NHWC(0, F, *d))
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()

    def timed(cmd, hint, ins, outs, reps=3):
        assert L.cmd_exec(cmd, hint, 0, ins, outs, s) == 0
        L.dll.nnc_mi355x_event_record(e0, s)
        for _ in range(reps):
            L.cmd_exec(cmd, hint, 0, ins, outs, s)
        L.dll.nnc_mi355x_event_record(e1, s)
        return L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps

    print("batch %d; filter gradient + bias gradient, ms per command; [..] = Winograd-domain MFMA TFLOP/s (2 x 36 x tiles x K x C)" % batch)
    for hw, c, k in LAYERS:
        a, w, g, dw, db = mk(batch, hw, hw, c), mk(k, 3, 3, c), mk(batch, hw, hw, k), mk(k, 3, 3, c), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, g], s)
        hint = nnc.HINT((1, 1), (1, 1))
        tiles = batch * ((hw + 3) // 4) ** 2
        wflops = 2.0 * 36 * tiles * k * c
        out = []
        for algo in (1, 2):
            if algo == 2 and (c % 64 or k % 32):
                continue
            cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c)
            cmd.algorithm = algo
            ms = timed(cmd, hint, [g, a, w], [None, dw, db])
            out.append("a%d %8.3f ms [%5.1f] (%s)" % (algo, ms, wflops / (ms * 1e-3) / 1e12, L.dll.nnc_mi355x_last_kernel_name().decode()))
        print("%-18s " % ("%d, %d->%d" % (hw, c, k)) + "   ".join(out), flush=True)
        for t in (a, w, g, dw, db):
            t.free()


if __name__ == "__main__":
    main()
