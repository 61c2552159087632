#!/usr/bin/env python
"""nnc_mi355x_depalettize on the MI355X (ccv_amd/csrc/palette.cpp), HIP-event timed: GB/s of algorithmic traffic -- the palettized byte stream read once
(palettes + q / 8 bytes per element) and the dense tensor written once -- against the 8 TB/s HBM peak, for the shapes palettized weights come in
(a 4096 x 4096 and a 8192 x 28672 half-precision matrix, 4 / 6 / 8 bits, blocks of 128 ... 16384 elements).
usage: python tools/palette_bench.py > gpurun_out/palette_bench.txt      (dry run on the emulator: PALETTE_BENCH_LIB=tests/emu/_build/libnnc_mi355x_emu.so PALETTE_BENCH_TINY=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccv_amd import nnc

L = nnc.load(os.environ.get("PALETTE_BENCH_LIB"))
s = L.stream_new(0)
e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
tiny = bool(os.environ.get("PALETTE_BENCH_TINY"))
ES = {nnc.CCV_16F: 2, nnc.CCV_32F: 4}
NAMES = {nnc.CCV_16F: "16F", nnc.CCV_32F: "32F"}


def timed(fn, reps):
    fn()
    L.dll.nnc_mi355x_event_record(e0, s)
    for _ in range(reps):
        fn()
    L.dll.nnc_mi355x_event_record(e1, s)
    L.stream_wait(s)
    return L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps


counts = [4096 * 33] if tiny else [4096 * 4096, 8192 * 28672]
for count in counts:
    for datatype in (nnc.CCV_16F, nnc.CCV_32F):
        for qbits, nib in ((4, 128), (4, 2048), (6, 512), (6, 4096), (8, 1280), (8, 16384)):
            nbytes = L.palettized_bytes(datatype, count, qbits, nib)
            rng = np.random.default_rng(0)
            src = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_8U, (nbytes + 64,), 0))
            block = rng.integers(0, 256, size=1 << 22, dtype=np.uint8)  # any bytes are a valid stream: every index is < 2^q, every palette word is moved as is
            src.upload(np.resize(block, nbytes + 64))
            dst = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, datatype, (count,), 0))
            ms = timed(lambda: L.depalettize(src, datatype, nbytes, qbits, nib, dst, count, s) == 0 or sys.exit("refused"), 2 if tiny else 20)
            kernel = L.dll.nnc_mi355x_last_kernel_name().decode()
            total = nbytes + count * ES[datatype]
            print("%11d x %s, %d bits, %5d per block: %8.3f ms  %7.1f GB/s algorithmic (%5.1f MB in, %6.1f MB out)  %.3f of 8 TB/s  [%s]"
                  % (count, NAMES[datatype], qbits, nib, ms, total / ms / 1e6, nbytes / 1e6, count * ES[datatype] / 1e6, total / ms / 1e6 / 8000.0, kernel), flush=True)
            src.free()
            dst.free()
