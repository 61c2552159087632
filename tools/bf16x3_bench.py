#!/usr/bin/env python
"""The Winograd-domain contractions (and the fc layers' GEMMs) of the VGG-D step on the bf16 matrix pipe with exactly split fp32 operands
(mfma_gemm_bf16x3.h) next to the fp32 matrix instructions: per-launch time / fp32-equivalent TFLOP/s from the backend's HIP-event records, and
the results against each other and against float64 (numpy) -- the 1e-4 relative bound of tests/test_parity_fullsize.py, unchanged.
usage: tools/bf16x3_bench.py > gpurun_out/bf16x3_bench.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ccv_amd import nnc

L = nnc.load()
MODES = [int(x) for x in os.environ.get("BF16X3_MODES", "0,3,4").split(",")]


def tens(*dims, fill=None, seed=0):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, dims, 0))
    if fill is not None:
        rng = np.random.default_rng(seed)
        n = int(np.prod(dims))
        block = ((rng.random(1 << 20, dtype=np.float32) - 0.5) * fill).astype(np.float32)
        t.upload(np.resize(block, n).reshape(dims))
    return t


def run(label, cmd, hint, ins, outs, reps=4):
    for _ in range(2):
        assert L.cmd_exec(cmd, hint, 0, ins, outs) == 0
    L.stream_wait(None)
    L.profile_enable(1)
    for _ in range(reps):
        assert L.cmd_exec(cmd, hint, 0, ins, outs) == 0
    L.stream_wait(None)
    recs = L.profile_records()
    L.profile_enable(0)
    by = {}
    for name, fl, _b, ms, dims in recs:
        k = by.setdefault(name.split("|")[0] + " " + name.split("|")[1][-44:] + " " + str(tuple(dims)), [0.0, 0.0])
        k[0] += fl; k[1] += ms
    for k, (fl, ms) in by.items():
        print("%-30s %-100s %8.3f ms/launch %8.1f TFLOP/s" % (label, k, ms / reps, fl / (ms * 1e-3) / 1e12))
    sys.stdout.flush()


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(1e-30, np.abs(b.astype(np.float64)).max()))


# 1. plain GEMMs against float64
for m, n, k in ((512, 384, 1024), (4096, 4096, 4096), (256, 4096, 18432)):
    a, w, b = tens(m, k, fill=1.0, seed=1), tens(n, k, fill=0.05, seed=2), tens(m, n)
    want = None
    if m * n * k <= 512 * 384 * 1024:
        want = a.numpy().astype(np.float64) @ w.numpy().astype(np.float64).T
    base = None
    for mode in MODES:
        L.tune_set("GEMM_BF16X3", mode)
        run("gemm %dx%dx%d mode %d" % (m, n, k, mode), nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [a, w], [b])
        got = b.numpy()
        if base is None:
            base = got
        print("    mode %d: max |diff| / max |ref| vs mode %d = %.3e%s" % (mode, MODES[0], rel(got, base), "" if want is None else "; vs float64 = %.3e" % rel(got, want)))
    del a, w, b

# 1b. the same GEMM on all-zero operands: the matrix pipe's rate when no operand bit toggles (is the kernel bound by its instruction stream or by the chip's power?)
if os.environ.get("BF16X3_ZERO", "1") == "1":
    m = n = k = 4096
    a, w, b = tens(m, k, fill=0.0), tens(n, k, fill=0.0), tens(m, n)
    for mode in MODES:
        L.tune_set("GEMM_BF16X3", mode)
        run("gemm zeros %dx%dx%d mode %d" % (m, n, k, mode), nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [a, w], [b])
    del a, w, b

# 1c. the fc layers' backward GEMMs (mixed operand layouts)
for m, n, k in ((256, 4096, 18432), (256, 4096, 4096)):
    a, w, g = tens(m, k, fill=1.0, seed=1), tens(n, k, fill=0.05, seed=2), tens(m, n, fill=0.1, seed=7)
    h, dw, db = tens(m, k), tens(n, k), tens(n)
    base = None
    for mode in MODES:
        L.tune_set("GEMM_BF16X3", mode)
        run("gemm bwd %dx%dx%d mode %d" % (m, n, k, mode), nnc.CMD_GEMM_BACKWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [g, a, w], [h, dw, db])
        got = (h.numpy(), dw.numpy())
        if base is None:
            base = got
        print("    mode %d vs mode %d: dx %.3e dw %.3e" % ((mode, MODES[0]) + tuple(rel(x, y) for x, y in zip(got, base))))
    del a, w, g, h, dw, db

# 2. the 3 x 3 layers of VGG-D whose Winograd-domain contractions go through HBM (algorithm 1), forward and backward
ALGO_WINO = int(os.environ.get("BF16X3_ALGO", "1"))
for nb, hw, c, kk in ((256, 28, 512, 512), (256, 14, 512, 512), (256, 56, 256, 256), (256, 28, 256, 512)):
    a, w, bias, b = tens(nb, hw, hw, c, fill=1.0, seed=3), tens(kk, 3, 3, c, fill=0.05, seed=4), tens(kk, fill=0.1, seed=5), tens(nb, hw, hw, kk)
    g, h, dw, db = tens(nb, hw, hw, kk, fill=0.1, seed=6), tens(nb, hw, hw, c), tens(kk, 3, 3, c), tens(kk)
    hint = nnc.HINT((1, 1), (1, 1))
    base = None
    for mode in MODES:
        L.tune_set("GEMM_BF16X3", mode)
        cmd = nnc.CMD_CONVOLUTION_FORWARD(1, kk, 3, 3, c); cmd.algorithm = ALGO_WINO
        run("conv fwd %dx%d^2x%d->%d mode %d" % (nb, hw, c, kk, mode), cmd, hint, [a, w, bias], [b])
        cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, kk, 3, 3, c); cmd.algorithm = ALGO_WINO
        run("conv bwd %dx%d^2x%d->%d mode %d" % (nb, hw, c, kk, mode), cmd, hint, [g, a, w], [h, dw, db])
        got = (b.numpy(), h.numpy(), dw.numpy())
        if base is None:
            base = got
        print("    mode %d vs mode %d: fwd %.3e dgrad %.3e wgrad %.3e" % ((mode, MODES[0]) + tuple(rel(x, y) for x, y in zip(got, base))))
    del a, w, bias, b, g, h, dw, db
L.tune_set("GEMM_BF16X3", 0)
