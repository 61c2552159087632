#!/usr/bin/env python
"""Scaled-dot-product attention forward AND backward on the MI355X: the matrix-core kernels (sdpa_forw_mfma_kernel, sdpa_dq_mfma_kernel, sdpa_dkv_mfma_kernel: fp32 MFMA)
against the VALU kernels of cmd_attention.cpp on the same tensors (tuning key SDPA_MFMA), HIP-event timed.  The backward command includes its own forward
re-run (output + log-sum-exp into scratch) and the delta pass: 3.5 x the forward's products in all.  usage: python tools/attention_bench.py > gpurun_out/attention_bench.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ccv_amd import nnc
import test_attention as T

L = nnc.load()
s = L.stream_new(0)
e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
F = nnc.CCV_32F


def tens(*dims, fill=0.5, dtype=F):
    t = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, dtype, dims, 0))
    rng = np.random.default_rng(0)
    block = (rng.random(1 << 20, dtype=np.float32) - 0.5) * fill
    t.upload(np.resize(block, int(np.prod(dims))).reshape(dims).astype(np.float16 if dtype == nnc.CCV_16F else np.float32))
    return t


for (B, R, C, H, Hk, D, causal) in [(8, 2048, 2048, 16, 16, 64, False), (8, 2048, 2048, 16, 16, 64, True), (4, 4096, 4096, 16, 4, 128, True), (32, 512, 512, 12, 12, 64, False)]:
    q, k, v, o = tens(B, R, H, D), tens(B, C, Hk, D), tens(B, C, Hk, D), tens(B, R, H, D)
    cmd = T.sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD", float(1.0 / np.sqrt(D)), causal)
    flops = 2.0 * B * H * R * C * 2 * D * (0.5 if causal else 1.0)
    line = "B %d R %d C %d heads %d/%d D %d %s:" % (B, R, C, H, Hk, D, "causal" if causal else "full  ")
    for mode in (1, 0):
        L.tune_set("SDPA_MFMA", mode)
        reps = 3 if mode else 1
        assert L.cmd_exec(cmd, nnc.NO_HINT, 0, [q, k, v], [o], s) == 0
        L.dll.nnc_mi355x_event_record(e0, s)
        for _ in range(reps):
            assert L.cmd_exec(cmd, nnc.NO_HINT, 0, [q, k, v], [o], s) == 0
        L.dll.nnc_mi355x_event_record(e1, s)
        ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
        line += "   %s %8.3f ms %6.1f TFLOP/s" % ("matrix cores" if mode else "VALU kernel ", ms, flops / ms / 1e9)
    L.tune_set("SDPA_MFMA", 1)
    print(line, flush=True)
    # the same forward on CCV_16F tensors: the f16 matrix-core kernel (sdpa_forw_f16_kernel)
    qh, kh, vh, oh = (tens(*d, dtype=nnc.CCV_16F) for d in ((B, R, H, D), (B, C, Hk, D), (B, C, Hk, D), (B, R, H, D)))
    assert L.cmd_exec(cmd, nnc.NO_HINT, 0, [qh, kh, vh], [oh], s) == 0
    L.dll.nnc_mi355x_event_record(e0, s)
    for _ in range(5):
        assert L.cmd_exec(cmd, nnc.NO_HINT, 0, [qh, kh, vh], [oh], s) == 0
    L.dll.nnc_mi355x_event_record(e1, s)
    L.stream_wait(s)
    ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / 5
    print("   forward in half precision (f16 matrix cores)                 %8.3f ms %6.1f TFLOP/s" % (ms, flops / ms / 1e9), flush=True)
    gh, dqh, dkh, dvh = (tens(*d, dtype=nnc.CCV_16F) for d in ((B, R, H, D), (B, R, H, D), (B, C, Hk, D), (B, C, Hk, D)))
    bcmd_h = T.sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_BACKWARD", float(1.0 / np.sqrt(D)), causal)
    assert L.cmd_exec(bcmd_h, nnc.NO_HINT, 0, [gh, None, None, qh, kh, vh], [dqh, dkh, dvh], s) == 0
    L.dll.nnc_mi355x_event_record(e0, s)
    for _ in range(3):
        assert L.cmd_exec(bcmd_h, nnc.NO_HINT, 0, [gh, None, None, qh, kh, vh], [dqh, dkh, dvh], s) == 0
    L.dll.nnc_mi355x_event_record(e1, s)
    L.stream_wait(s)
    ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / 3
    print("   backward in half precision (f16 matrix cores, 7 products)    %8.3f ms %6.1f TFLOP/s" % (ms, 3.5 * flops / ms / 1e9), flush=True)
    for t in (qh, kh, vh, oh, gh, dqh, dkh, dvh):
        t.free()
    g, dq, dk, dv = tens(B, R, H, D), tens(B, R, H, D), tens(B, C, Hk, D), tens(B, C, Hk, D)
    bcmd = T.sdpa_cmd("SCALED_DOT_PRODUCT_ATTENTION_BACKWARD", float(1.0 / np.sqrt(D)), causal)
    line = "   backward (forward re-run + dq + dk / dv: 7 products)        "
    for mode in (1, 0):
        L.tune_set("SDPA_MFMA", mode)
        reps = 3 if mode else 1
        assert L.cmd_exec(bcmd, nnc.NO_HINT, 0, [g, None, None, q, k, v], [dq, dk, dv], s) == 0
        L.dll.nnc_mi355x_event_record(e0, s)
        for _ in range(reps):
            assert L.cmd_exec(bcmd, nnc.NO_HINT, 0, [g, None, None, q, k, v], [dq, dk, dv], s) == 0
        L.dll.nnc_mi355x_event_record(e1, s)
        L.stream_wait(s)
        ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
        line += "   %s %8.3f ms %6.1f TFLOP/s" % ("matrix cores" if mode else "VALU kernels", ms, 3.5 * flops / ms / 1e9)
    L.tune_set("SDPA_MFMA", 1)
    print(line, flush=True)
    for t in (q, k, v, o, g, dq, dk, dv):
        t.free()
