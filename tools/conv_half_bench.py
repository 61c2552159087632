#!/usr/bin/env python
"""Time the 3 x 3 half-precision NCHW convolutions of the CIFAR-10 DawnNet (batch 512) and ResNet-50 (batch 256) -- forward, data gradient, filter gradient -- by
the backend's per-launch records (the contraction kernel alone, without the layout passes around it), under the launcher's own choice and under forced block tiles /
K-slice counts (nnc_mi355x_debug_force_tile / _force_splits).  Feeds the tile / split rules of gemm_launch.h for mfma_gemm_f16_kernel.
usage: python tools/conv_half_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccv_amd import nnc

# (batch, C_in, C_out, H = W)
SHAPES = [(512, 64, 128, 32), (512, 128, 128, 16), (512, 128, 256, 16), (512, 256, 512, 8), (512, 512, 512, 4),
          (256, 64, 64, 56), (256, 128, 128, 28), (256, 256, 256, 14), (256, 512, 512, 7)]
MODES = [((0, 0), 0), ((1, 1), 0), ((4, 2), 0), ((2, 2), 8), ((1, 1), 8)]


def main():
    L = nnc.load()
    s = L.stream_new(0)
    F = nnc.CCV_16F
    mk = lambda *d: L.tensor(nnc.GPU_TENSOR_NCHW(0, F, *d))
    print("ms of the contraction kernel [TFLOP/s] per (tile, forced K-slices); (0, 0) / 0 = the launcher's choice")
    for n, c, k, hw in SHAPES:
        a, w, b, bias = mk(n, c, hw, hw), mk(k, c, 3, 3), mk(n, k, hw, hw), mk(k)
        g, h, dw, db = mk(n, k, hw, hw), mk(n, c, hw, hw), mk(k, c, 3, 3), mk(k)
        L.cmd_exec(nnc.CMD_SET_FORWARD(0.01), nnc.HINT(), 0, [], [a, w, bias, g], s)
        hint = nnc.HINT((1, 1), (1, 1))
        flops = 2.0 * n * hw * hw * k * c * 9
        out = {"conv_fwd_h": [], "conv_dgrad_h": [], "conv_wgrad_h": []}
        for (wm, wn), sp in MODES:
            L.force_tile(wm, wn); L.force_splits(sp)
            for _ in range(2):  # second pass is the one recorded
                L.profile_enable(0); L.profile_enable(1)
                L.cmd_exec(nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), hint, 0, [a, w, bias], [b], s)
                L.cmd_exec(nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c), hint, 0, [g, a, w], [h, dw, db], s)
                L.stream_wait(s)
            recs = L.profile_records()
            for name in out:
                ms = sum(r[3] for r in recs if r[0].startswith(name + "|"))
                out[name].append("(%d,%d)/%-2d %6.3f [%5.1f]" % (wm, wn, sp, ms, flops / (ms * 1e-3) / 1e12 if ms > 0 else 0))
        L.force_tile(0, 0); L.force_splits(0); L.profile_enable(0)
        for name in out:
            print("%-22s %-13s " % ("%d x %d->%d @%d^2" % (n, c, k, hw), name) + "  ".join(out[name]), flush=True)
        for t in (a, w, b, bias, g, h, dw, db):
            t.free()


if __name__ == "__main__":
    main()
