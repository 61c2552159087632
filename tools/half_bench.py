#!/usr/bin/env python
"""Half-precision contraction core on the MI355X (mfma_gemm_f16.h): achieved TFLOP/s of GEMM and convolution launches, CCV_16F
next to CCV_32F on the same shapes, from the backend's HIP-event launch records.  usage: tools/half_bench.py > gpurun_out/half_bench.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ccv_amd import nnc

L = nnc.load()
DT = {"f32": (nnc.CCV_32F, np.float32), "f16": (nnc.CCV_16F, np.float16)}


def tens(dt, *dims, fill=None):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, DT[dt][0], dims, 0))
    if fill is not None:
        rng = np.random.default_rng(0)
        n = int(np.prod(dims))
        block = ((rng.random(1 << 20, dtype=np.float32) - 0.5) * fill).astype(DT[dt][1])
        t.upload(np.resize(block, n).reshape(dims))
    return t


def run(label, cmd, hint, ins, outs, reps=5):
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
        k = by.setdefault(name.split("|")[0] + " " + name.split("|")[1][-48:], [0.0, 0.0])
        k[0] += fl; k[1] += ms
    for k, (fl, ms) in by.items():
        print("%-34s %-70s %8.3f ms/launch %8.1f TFLOP/s" % (label, k, ms / reps, fl / (ms * 1e-3) / 1e12))
    sys.stdout.flush()


for dt in ("f32", "f16"):
    for m, n, k in ((4096, 4096, 4096), (8192, 8192, 8192), (256, 4096, 18432)):
        a, w, b = tens(dt, m, k, fill=1.0), tens(dt, n, k, fill=0.05), tens(dt, m, n)
        run("gemm %s %dx%dx%d" % (dt, m, n, k), nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), nnc.NO_HINT, [a, w], [b])
        del a, w, b
    for nb, hw, c, kk in ((256, 56, 256, 256), (256, 28, 512, 512), (512, 32, 128, 128)):
        a, w, bias, b = tens(dt, nb, hw, hw, c, fill=1.0), tens(dt, kk, 3, 3, c, fill=0.05), tens(dt, kk, fill=0.1), tens(dt, nb, hw, hw, kk)
        hint = nnc.HINT((1, 1), (1, 1))
        cmd = nnc.CMD_CONVOLUTION_FORWARD(1, kk, 3, 3, c)
        if dt == "f32":
            cmd.algorithm = 0  # the implicit GEMM, for a like-for-like kernel comparison (the backend's own choice is Winograd)
        run("conv fwd %s %dx%dx%dx%d->%d" % (dt, nb, hw, hw, c, kk), cmd, hint, [a, w, bias], [b])
        g, h, dw, db = tens(dt, nb, hw, hw, kk, fill=0.1), tens(dt, nb, hw, hw, c), tens(dt, kk, 3, 3, c), tens(dt, kk)
        cmd = nnc.CMD_CONVOLUTION_BACKWARD(1, kk, 3, 3, c)
        if dt == "f32":
            cmd.algorithm = 0
        run("conv bwd %s %dx%dx%dx%d->%d" % (dt, nb, hw, hw, c, kk), cmd, hint, [g, a, w], [h, dw, db])
        del a, w, bias, b, g, h, dw, db
