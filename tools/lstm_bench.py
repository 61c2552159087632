#!/usr/bin/env python
"""LSTM forward / backward on the MI355X (cmd_lstm.cpp), HIP-event timed, at the shapes of the reference's recurrent trainer (test/int/nnc/imdb.tests.c:1278: batch 64,
sequences of up to 512 steps, 128 features) and a wider one.  Algorithmic flops: 2 * T * B * 4H * (in + P) per pseudo-layer forward, three times that backward.
usage: python tools/lstm_bench.py > gpurun_out/lstm_bench.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccv_amd import nnc

L = nnc.load(os.environ.get("LSTM_BENCH_LIB"))  # (a dry run of this script on the emulator: LSTM_BENCH_LIB=tests/emu/_build/libnnc_mi355x_emu.so LSTM_BENCH_TINY=1)
s = L.stream_new(0)
e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()


def tens(*dims, scale=0.2):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, dims, 0))
    rng = np.random.default_rng(0)
    block = (rng.random(1 << 20, dtype=np.float32) - 0.5) * scale
    t.upload(np.resize(block, int(np.prod(dims))).reshape(dims))
    return t


def timed(fn, reps):
    fn()
    L.dll.nnc_mi355x_event_record(e0, s)
    for _ in range(reps):
        fn()
    L.dll.nnc_mi355x_event_record(e1, s)
    L.stream_wait(s)
    return L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps


SHAPES = [(4, 3, 8, 8, 2, 1)] if os.environ.get("LSTM_BENCH_TINY") else [(512, 64, 128, 128, 1, 0), (128, 64, 128, 128, 2, 1), (256, 128, 512, 512, 1, 0)]
for (T, B, I, H, layers, bidir) in SHAPES:
    D = 2 if bidir else 1
    nw = sum(D * (4 * H * (I if l == 0 else D * H) + 4 * H * H) for l in range(layers)) + layers * D * 8 * H
    fcmd = nnc.CMD_LSTM_FORWARD(H, 0, layers, 1, 0, bidir, 0.0, 0)
    bcmd = nnc.CMD_LSTM_BACKWARD(H, 0, layers, 1, 0, bidir, 0.0, 0)
    rrows = (L.dll.nnc_mi355x_lstm_reserve_space_size(fcmd, nnc.CCV_32F, I, B, T) // 4 + H - 1) // H
    x, w, y, r = tens(T, B, I), tens(nw // H, H), tens(T, B, D * H), tens(rrows, H)
    hx, cx, hy, cy = (tens(layers * D, B, H) for _ in range(4))
    dy, dx, dw, dhx, dcx = tens(T, B, D * H), tens(T, B, I), tens(nw // H, H), tens(layers * D, B, H), tens(layers * D, B, H)
    flops = sum(2.0 * T * B * 4 * H * ((I if l == 0 else D * H) + H) * D for l in range(layers))
    ok = lambda r: r == 0 or sys.exit("command returned %d" % r)
    steps = T * layers * D
    for mode, rows in ((1, 1), (1, 2), (1, 0), (0, 0)):  # a batch row per workgroup (H <= 128) / two rows per workgroup; hidden units per workgroup with a hand-over per step; a launch per step
        L.tune_set("LSTM_PERSISTENT", mode)
        L.tune_set("LSTM_ROWS", rows)
        f_ms = timed(lambda: ok(L.cmd_exec(fcmd, nnc.NO_HINT, 0, [x, None, hx, cx, w], [y, hy, cy, r], s)), 3)
        fk = L.dll.nnc_mi355x_last_kernel_name().decode()
        b_ms = timed(lambda: ok(L.cmd_exec(bcmd, nnc.NO_HINT, 0, [dy, None, None, None, x, None, hx, cx, w, y, hy, cy, r], [dx, None, dhx, dcx, dw], s)), 3)
        bk = L.dll.nnc_mi355x_last_kernel_name().decode()
        print("T %4d B %4d in %4d hidden %4d layers %d %s: forward %8.3f ms (%5.1f us per step, %6.2f TFLOP/s, %s)   backward %8.3f ms (%5.1f us per step, %6.2f TFLOP/s, %s)"
              % (T, B, I, H, layers, "both directions" if bidir else "one direction  ", f_ms, f_ms * 1e3 / steps, flops / f_ms / 1e9, fk, b_ms, b_ms * 1e3 / steps, 3 * flops / b_ms / 1e9, bk), flush=True)
    L.tune_set("LSTM_PERSISTENT", 1)
    L.tune_set("LSTM_ROWS", 1)
