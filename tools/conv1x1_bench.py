#!/usr/bin/env python
"""Time the 1x1 NCHW convolutions of ResNet-50 v1d (batch 256) -- forward, data gradient, filter gradient -- under the launcher's own tile choice and
under each forced block tile (nnc_mi355x_debug_force_tile), HIP-event timed.  Feeds gemm_pick_tile() / the small-K rules of gemm_launch.h.
usage: python tools/conv1x1_bench.py [batch] [f32|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccv_amd import nnc

# (C_in, C_out, H = W)
SHAPES = [(64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28), (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (2048, 512, 7), (64, 64, 56)]
TILES = [(0, 0), (2, 2), (2, 1), (1, 2), (1, 1)]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    half = len(sys.argv) > 2 and sys.argv[2] == "f16"
    L = nnc.load()
    s = L.stream_new(0)
    F = nnc.CCV_16F if half else nnc.CCV_32F
    es = 2 if half else 4
    mk = lambda *d: L.tensor(nnc.GPU_TENSOR_NCHW(0, F, *d))
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()

    def timed(cmd, hint, ins, outs, reps=5):
        L.cmd_exec(cmd, hint, 0, ins, outs, s)
        L.dll.nnc_mi355x_event_record(e0, s)
        for _ in range(reps):
            L.cmd_exec(cmd, hint, 0, ins, outs, s)
        L.dll.nnc_mi355x_event_record(e1, s)
        L.stream_wait(s)
        return L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps

    print("batch %d %s; per pass: ms [TFLOP/s, TB/s of the algorithmic bytes] under tile (WM, WN); (0, 0) = the launcher's choice" % (batch, "f16" if half else "f32"))
    for c, k, hw in SHAPES:
        a, w, b, bias = mk(batch, c, hw, hw), mk(k, c, 1, 1), mk(batch, k, hw, hw), mk(k)
        g, h, dw, db = mk(batch, k, hw, hw), mk(batch, c, hw, hw), mk(k, c, 1, 1), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, bias, g], s)
        hint = nnc.HINT((1, 1), (0, 0))
        flops = 2.0 * batch * hw * hw * k * c
        na, nb = batch * c * hw * hw * es, batch * k * hw * hw * es
        for what in ("fwd", "dgrad", "wgrad"):
            out = []
            for wm, wn in TILES:
                L.force_tile(wm, wn)
                if what == "fwd":
                    ms = timed(nnc.CMD_CONVOLUTION_FORWARD(1, k, 1, 1, c), hint, [a, w, bias], [b])
                elif what == "dgrad":
                    ms = timed(nnc.CMD_CONVOLUTION_BACKWARD(1, k, 1, 1, c), hint, [g, None, w], [h])
                else:
                    ms = timed(nnc.CMD_CONVOLUTION_BACKWARD(1, k, 1, 1, c), hint, [g, a, None], [None, dw, db])
                out.append("(%d,%d) %6.3f [%5.1f %4.2f]" % (wm, wn, ms, flops / (ms * 1e-3) / 1e12, (na + nb) / (ms * 1e-3) / 1e12))
            L.force_tile(0, 0)
            print("%-16s %-5s " % ("%d->%d @%d^2" % (c, k, hw), what) + "  ".join(out), flush=True)
        for t in (a, w, b, bias, g, h, dw, db):
            t.free()


if __name__ == "__main__":
    main()
