"""Palettized tensors (CCV_QX) on the backend: nnc_mi355x_depalettize (ccv_amd/csrc/palette.cpp; replaces ccv_nnc_compat_depalettize,
lib/nnc/gpu/ccv_nnc_palettize.cu) and the rows that take palettized inputs (GEMM, convolution, DATA_TRANSFER).

The checker is the reference itself: its quantiser `ccv_nnc_palettize` writes the byte stream and its CPU reader `ccv_nnc_depalettize`
(lib/nnc/ccv_nnc_palettize.c:9-208, 211-956, compiled into oracle/_ref/libccv_ref.so) says what the stream means; `pack_stream` (oracle/palettize_numpy.py) restates the layout in
numpy (any palette, any indices -- the quantiser only ever produces k-means palettes) and is pinned to both.  Results are moved bits: every comparison is exact."""
import ctypes as C
import numpy as np
import pytest
from ccv_amd import nnc
from harness import make_tensors

import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from palettize_numpy import WORD as _NP, index_bytes_per_block, pack_stream, expand  # noqa: E402  (the numpy restatement of the stream's layout, pinned below)


def random_case(rng, count, qbits, nib, datatype):
    blocks = (count + nib - 1) // nib
    info = np.iinfo(_NP[datatype])
    palettes = rng.integers(0, info.max, size=(blocks, 1 << qbits), dtype=_NP[datatype], endpoint=True)
    indices = rng.integers(0, 1 << qbits, size=count)
    return palettes, indices


def device_bytes(lib, stream_bytes):
    (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.concatenate([stream_bytes, np.zeros(16, np.uint8)])])
    return t


# ---- the restatement of the layout is the reference's -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("datatype", [nnc.CCV_16F, nnc.CCV_32F, nnc.CCV_64F])
@pytest.mark.parametrize("qbits,nib,count", [(4, 128, 2839), (5, 128, 2839), (6, 128, 2839), (7, 128, 2839), (8, 128, 2839), (6, 512, 2840), (7, 512, 2840), (8, 1280, 8192), (5, 8, 13), (4, 2, 7)])
def test_the_numpy_stream_is_what_the_reference_reader_expands(ref_lib, datatype, qbits, nib, count):
    rng = np.random.default_rng(qbits * 1000 + nib)
    palettes, indices = random_case(rng, count, qbits, nib, datatype)
    if datatype != nnc.CCV_64F:  # the reader copies 16F / 32F as integers; doubles go through `double` loads: keep them finite
        pass
    else:
        palettes = rng.standard_normal(palettes.shape).view(np.uint64)
    stream = pack_stream(palettes, indices, qbits, nib, datatype)
    out = np.zeros(count, _NP[datatype])
    d = ref_lib.dll
    d.ccv_nnc_depalettize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    d.ccv_nnc_depalettize.restype = None
    buf = np.concatenate([stream, np.zeros(64, np.uint8)])  # (the CPU reader touches whole groups past a short tail)
    d.ccv_nnc_depalettize(buf.ctypes.data, datatype, nnc.CPU_MEMORY, stream.nbytes, qbits, nib, out.ctypes.data, count)
    assert (out == expand(palettes, indices, nib, datatype)).all()


@pytest.mark.parametrize("datatype,np_t", [(nnc.CCV_32F, np.float32), (nnc.CCV_64F, np.float64)])
@pytest.mark.parametrize("qbits,nib", [(4, 128), (5, 128), (6, 512), (7, 512), (8, 1280)])
def test_the_reference_quantiser_writes_the_numpy_stream(ref_lib, datatype, np_t, qbits, nib):
    """Values drawn from exactly 2^qbits levels per block quantise losslessly (the reference's own palettize cases, test/int/nnc/palettize.tests.c): the stream
    the quantiser writes expands -- by the numpy restatement of the layout -- to the values that went in, and has the size the formula says."""
    count = 2839
    levels = np.arange(1, (1 << qbits) + 1, dtype=np_t)
    values = levels[np.arange(count) % (1 << qbits)]
    d = ref_lib.dll
    d.ccv_nnc_palettize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    d.ccv_nnc_palettize.restype = C.c_size_t
    blocks = (count + nib - 1) // nib
    cap = blocks * ((1 << qbits) * values.itemsize + index_bytes_per_block(qbits, nib)) + 64
    stream = np.zeros(cap, np.uint8)
    n = d.ccv_nnc_palettize(values.ctypes.data, datatype, nnc.CPU_MEMORY, count, qbits, nib, stream.ctypes.data, cap)
    assert 0 < n <= cap
    # read the stream back with the layout of pack_stream: palette words + MSB-first indices
    word = np.dtype(_NP[datatype])
    stride = (1 << qbits) * word.itemsize + index_bytes_per_block(qbits, nib)
    got = np.zeros(count, word)
    for b in range(blocks):
        base = b * stride
        pal = stream[base:base + (1 << qbits) * word.itemsize].view(word)
        m = min(nib, count - b * nib)
        bits = np.unpackbits(stream[base + (1 << qbits) * word.itemsize: base + (1 << qbits) * word.itemsize + (m * qbits + 7) // 8])[:m * qbits].reshape(m, qbits)
        idx = (bits.astype(np.int64) << np.arange(qbits - 1, -1, -1)).sum(axis=1)
        got[b * nib:b * nib + m] = pal[idx]
    assert (got == values.view(word)).all()


# ---- the kernels ----------------------------------------------------------------------------------------------------------------------------------------

DEPAL_CASES = [
    # qbits, elements per block, count -- both kernel forms (palette in LDS for blocks of 2048 elements and more with 4 uses per entry), full and ragged last blocks and groups
    (4, 128, 2839), (4, 128, 2840), (5, 128, 2839), (5, 128, 2840), (6, 128, 2839), (6, 512, 2840), (7, 128, 2839), (7, 512, 2840), (8, 128, 2839), (8, 1280, 8192),
    (4, 64, 64 * 40), (4, 4096, 4096 * 3 + 1), (5, 2048, 2048 * 2 + 1027), (6, 4096, 9000), (7, 8192, 8192 + 5), (8, 16384, 40000),
    (8, 8192 * 3, 8192 * 3 * 2), (4, 8, 5), (5, 8, 8), (6, 4, 10), (7, 16, 17), (8, 1, 9), (4, 2, 3), (6, 12, 31),
]


@pytest.mark.parametrize("datatype", [nnc.CCV_16F, nnc.CCV_32F, nnc.CCV_64F])
@pytest.mark.parametrize("qbits,nib,count", DEPAL_CASES)
def test_depalettize_expands_the_stream_bit_for_bit(backend, datatype, qbits, nib, count):
    lib = backend
    rng = np.random.default_rng(qbits * 7919 + nib * 31 + count)
    palettes, indices = random_case(rng, count, qbits, nib, datatype)
    stream = pack_stream(palettes, indices, qbits, nib, datatype)
    assert stream.nbytes >= lib.palettized_bytes(datatype, count, qbits, nib)
    src = device_bytes(lib, stream)
    sentinel = np.full(count + 16, 0xA5, np.uint8).astype(_NP[datatype]) * np.array(0x0101, _NP[datatype])
    (dst,) = make_tensors(lib, nnc.GPU_MEMORY, [sentinel.view({nnc.CCV_16F: np.float16, nnc.CCV_32F: np.float32, nnc.CCV_64F: np.float64}[datatype])])
    s = lib.stream_new(0)
    try:
        assert lib.depalettize(src, datatype, stream.nbytes, qbits, nib, dst, count, s) == 0
        lib.stream_wait(s)
    finally:
        lib.stream_free(s)
    got = dst.numpy().view(_NP[datatype])
    assert (got[:count] == expand(palettes, indices, nib, datatype)).all()
    assert (got[count:] == sentinel[count:]).all()  # nothing past the tensor's last element


@pytest.mark.parametrize("qbits,nib,count", [(5, 8, 13), (8, 16, 40), (4, 16, 64), (6, 4, 10), (7, 16, 17), (4, 128, 2839), (6, 4096, 9000)])
def test_depalettize_reads_nothing_past_the_stream(emu_lib, qbits, nib, count):
    """The stream of a tensor's last block ends with the last index byte its elements touch.  On the emulator device memory is host memory: the stream is placed
    so that it ends exactly at a page boundary with an inaccessible page behind it -- an 8-byte index load that reached past the stream's end (the kernel's fast
    path takes whole groups in one load) would fault."""
    import ctypes, mmap
    lib = emu_lib
    rng = np.random.default_rng(qbits + nib + count)
    palettes, indices = random_case(rng, count, qbits, nib, nnc.CCV_16F)
    stream = pack_stream(palettes, indices, qbits, nib, nnc.CCV_16F)
    n = int(lib.palettized_bytes(nnc.CCV_16F, count, qbits, nib))  # the tensor's own size: what a caller allocates (the quantiser may have written a longer tail)
    assert n <= stream.nbytes
    page = mmap.PAGESIZE
    pages = (n + 1 + page - 1) // page
    mm = mmap.mmap(-1, (pages + 1) * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    start = base + pages * page - n - (n & 1)  # (the palettes are read as 2-byte words: an odd-sized stream leaves one accessible byte behind it)
    ctypes.memmove(start, stream.ctypes.data, n)
    assert libc.mprotect(base + pages * page, page, 0) == 0  # PROT_NONE
    try:
        (dst,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(count, np.float16)])
        assert lib.depalettize(start, nnc.CCV_16F, n, qbits, nib, dst, count) == 0
        lib.stream_wait(None)
        assert (dst.numpy().view(np.uint16) == expand(palettes, indices, nib, nnc.CCV_16F)).all()
    finally:
        libc.mprotect(base + pages * page, page, mmap.PROT_READ | mmap.PROT_WRITE)


def test_depalettize_into_an_unaligned_destination(backend):
    """An output that does not start on 16 bytes (a tensor carved out of a larger allocation) takes the element-wise store path."""
    lib = backend
    count, qbits, nib = 4101, 6, 1024
    rng = np.random.default_rng(5)
    palettes, indices = random_case(rng, count, qbits, nib, nnc.CCV_16F)
    stream = pack_stream(palettes, indices, qbits, nib, nnc.CCV_16F)
    src = device_bytes(lib, stream)
    (dst,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(count + 8, np.float16)])
    assert lib.depalettize(src, nnc.CCV_16F, stream.nbytes, qbits, nib, dst.ptr + 2 * 3, count) == 0
    lib.stream_wait(None)
    got = dst.numpy().view(np.uint16)
    assert (got[3:3 + count] == expand(palettes, indices, nib, nnc.CCV_16F)).all() and (got[:3] == 0).all() and (got[3 + count:] == 0).all()


def test_depalettize_refuses_what_the_reference_asserts_on(backend):
    lib = backend
    (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(4096, np.uint8)])
    (o,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(64, np.float32)])
    assert lib.depalettize(t, nnc.CCV_32F, 4096, 3, 128, o, 64) == nnc.EXEC_INVALID   # 3-bit palettes do not exist
    assert lib.depalettize(t, nnc.CCV_32F, 4096, 9, 128, o, 64) == nnc.EXEC_INVALID
    assert lib.depalettize(t, nnc.CCV_32S, 4096, 4, 128, o, 64) == nnc.EXEC_INVALID   # palettes are 16F / 32F / 64F
    assert lib.depalettize(t, nnc.CCV_32F, 16, 4, 128, o, 64) == nnc.EXEC_INVALID     # a stream shorter than one palette + the indices
    assert lib.depalettize(t, nnc.CCV_32F, 4096, 4, 128, o, 0) == 0                   # nothing to do


# ---- rows with palettized inputs --------------------------------------------------------------------------------------------------------------------------

def lossless_stream(values, qbits, nib, datatype):
    """Quantise `values` (at most 2^qbits distinct words per block) the way the reference's quantiser would for such input: palette = the block's distinct words."""
    words = np.ascontiguousarray(values).reshape(-1).view(_NP[datatype])
    count = len(words)
    blocks = (count + nib - 1) // nib
    palettes = np.zeros((blocks, 1 << qbits), _NP[datatype])
    indices = np.zeros(count, np.int64)
    for b in range(blocks):
        u, inv = np.unique(words[b * nib:(b + 1) * nib], return_inverse=True)
        assert len(u) <= 1 << qbits
        palettes[b, :len(u)] = u
        indices[b * nib:b * nib + len(inv)] = inv
    return pack_stream(palettes, indices, qbits, nib, datatype)


def levels(rng, shape, qbits, dtype):
    return (rng.integers(0, 1 << qbits, size=shape) - (1 << (qbits - 1))).astype(dtype) / dtype(8)


@pytest.mark.parametrize("dtype,datatype", [(np.float32, nnc.CCV_32F), (np.float16, nnc.CCV_16F)])
@pytest.mark.parametrize("qbits,nib", [(4, 128), (6, 512), (8, 4096)])
def test_gemm_with_palettized_weights_equals_the_dense_command(backend, dtype, datatype, qbits, nib):
    """GEMM_FORWARD / GEMM_BACKWARD with a CCV_QX weight matrix (cublas.tests.c:212-330, 331-: `gemm no transpose with bias and palettize weights`, `backward
    gemm ... palettize weights`) are the dense commands on the expanded matrix, bit for bit -- and DATA_TRANSFER moves the byte stream (the case's first step)."""
    lib = backend
    rng = np.random.default_rng(qbits)
    a, w, bias = levels(rng, (24, 40), 4, dtype), levels(rng, (40, 56), qbits, dtype), levels(rng, (56,), 4, dtype)
    g = levels(rng, (24, 56), 4, dtype)
    stream = lossless_stream(w, qbits, nib, datatype)
    qparams = nnc.tensor_palettize(nnc.GPU_TENSOR_NHWC(0, datatype, 40, 56), qbits, nib)
    cparams = nnc.tensor_palettize(nnc.CPU_TENSOR_NHWC(datatype, 40, 56), qbits, nib)
    host_q = nnc.PalettizedTensor(lib, cparams, stream)
    dev_q = nnc.PalettizedTensor(lib, qparams, np.zeros_like(stream))
    assert lib.cmd_exec(nnc.CMD_DATA_TRANSFER_FORWARD(), nnc.NO_HINT, 0, [host_q], [dev_q]) == 0
    assert (dev_q.numpy() == stream).all()
    ta, tw, tb, tg = make_tensors(lib, nnc.GPU_MEMORY, [a, w, bias, g])
    (c_dense, c_pal) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros((24, 56), dtype)] * 2)
    assert lib.cmd_exec(nnc.CMD_GEMM_FORWARD(), nnc.NO_HINT, 0, [ta, tw, tb], [c_dense]) == 0
    assert lib.cmd_exec(nnc.CMD_GEMM_FORWARD(), nnc.NO_HINT, 0, [ta, dev_q, tb], [c_pal]) == 0
    assert (c_dense.numpy().view(_NP[datatype]) == c_pal.numpy().view(_NP[datatype])).all()
    assert np.abs(c_dense.numpy().astype(np.float64) - (a.astype(np.float64) @ w.astype(np.float64) + bias)).max() < (1e-3 if dtype == np.float32 else 0.5)
    # backward: the input gradient reads the palettized matrix, the weight gradient is a dense output as in the reference
    outs_d = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros((24, 40), dtype), np.zeros((40, 56), dtype), np.zeros((56,), dtype)])
    outs_p = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros((24, 40), dtype), np.zeros((40, 56), dtype), np.zeros((56,), dtype)])
    assert lib.cmd_exec(nnc.CMD_GEMM_BACKWARD(), nnc.NO_HINT, 0, [tg, ta, tw], outs_d) == 0
    assert lib.cmd_exec(nnc.CMD_GEMM_BACKWARD(), nnc.NO_HINT, 0, [tg, ta, dev_q], outs_p) == 0
    for x, y in zip(outs_d, outs_p):
        assert (x.numpy().view(_NP[datatype]) == y.numpy().view(_NP[datatype])).all()


@pytest.mark.parametrize("dtype,datatype,fmt", [(np.float32, nnc.CCV_32F, "NHWC"), (np.float16, nnc.CCV_16F, "NHWC"), (np.float16, nnc.CCV_16F, "NCHW")])
def test_convolution_with_palettized_filters_equals_the_dense_command(backend, dtype, datatype, fmt):
    """CONVOLUTION_FORWARD / BACKWARD with CCV_QX filters (cudnn.tests.c:212-: `cudnn forward convolution in half precision with palettize weights`; the data
    gradient's prologue ccv_nnc_conv_gpu_cudnn.cu:328-345)."""
    lib = backend
    rng = np.random.default_rng(3)
    N, H, W, Cc, K = 2, 9, 9, 8, 16
    if fmt == "NHWC":
        x, w, y = levels(rng, (N, H, W, Cc), 4, dtype), levels(rng, (K, 3, 3, Cc), 6, dtype), np.zeros((N, H, W, K), dtype)
    else:
        x, w, y = levels(rng, (N, Cc, H, W), 4, dtype), levels(rng, (K, Cc, 3, 3), 6, dtype), np.zeros((N, K, H, W), dtype)
    bias = levels(rng, (K,), 4, dtype)
    f = nnc.NHWC if fmt == "NHWC" else nnc.NCHW
    stream = lossless_stream(w, 6, 512, datatype)
    dev_q = nnc.PalettizedTensor(lib, nnc.tensor_palettize(nnc.tensor_param(nnc.GPU_MEMORY, f, datatype, w.shape, 0), 6, 512), stream)
    tx, tw, tb = make_tensors(lib, nnc.GPU_MEMORY, [x, w, bias], fmt)
    yd, yp = make_tensors(lib, nnc.GPU_MEMORY, [y, y], fmt)
    cmd = nnc.CMD_CONVOLUTION_FORWARD(1, K, 3, 3, Cc)
    hint = nnc.hint_auto(cmd, (H, W), (H, W))
    assert lib.cmd_exec(cmd, hint, 0, [tx, tw, tb], [yd]) == 0
    assert lib.cmd_exec(cmd, hint, 0, [tx, dev_q, tb], [yp]) == 0
    assert np.abs(yd.numpy().astype(np.float64)).max() > 0
    assert (yd.numpy().view(_NP[datatype]) == yp.numpy().view(_NP[datatype])).all()
    g = levels(rng, y.shape, 4, dtype)
    (tg,) = make_tensors(lib, nnc.GPU_MEMORY, [g], fmt)
    bcmd = nnc.CMD_CONVOLUTION_BACKWARD(1, K, 3, 3, Cc)
    outs_d = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros_like(x), np.zeros_like(w), np.zeros_like(bias)], fmt)
    outs_p = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros_like(x), np.zeros_like(w), np.zeros_like(bias)], fmt)
    assert lib.cmd_exec(bcmd, hint, 0, [tg, tx, tw], outs_d) == 0
    assert lib.cmd_exec(bcmd, hint, 0, [tg, tx, dev_q], outs_p) == 0
    for a, b in zip(outs_d, outs_p):
        assert (a.numpy().view(_NP[datatype]) == b.numpy().view(_NP[datatype])).all()


@pytest.mark.parametrize("dtype,datatype", [(np.float32, nnc.CCV_32F), (np.float16, nnc.CCV_16F)])
def test_transposed_convolution_with_palettized_filters_equals_the_dense_command(emu_lib, dtype, datatype):
    """CONVOLUTION_TRANSPOSE_FORWARD with CCV_QX filters (ccv_nnc_conv_transpose_gpu_cudnn.cu:72-90).  In half precision the row underneath is itself a wrapper
    (fp32 images of the half tensors, half_stage.cpp): the dense image of the filters sits in the palette arena while that wrapper grows its own.  CPU tier only
    (the GEMM / convolution / attention cases above run in both tiers)."""
    lib = emu_lib
    rng = np.random.default_rng(12)
    n, H, W, count, ca, k = 2, 9, 9, 16, 8, 3
    a, w, bias = levels(rng, (n, H, W, ca), 4, dtype), levels(rng, (ca, k, k, count), 5, dtype), levels(rng, (count,), 4, dtype)
    cmd = nnc.CMD_CONVOLUTION_TRANSPOSE_FORWARD(1, count, 0, k, k, ca)
    hint = nnc.HINT((1, 1), (1, 1))
    dev_q = nnc.PalettizedTensor(lib, nnc.tensor_palettize(nnc.GPU_TENSOR_NHWC(0, datatype, ca, k, k, count), 5, 128), lossless_stream(w, 5, 128, datatype))
    ta, tw, tb = make_tensors(lib, nnc.GPU_MEMORY, [a, w, bias])
    od, op = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros((n, H, W, count), dtype)] * 2)
    assert lib.cmd_exec(cmd, hint, 0, [ta, tw, tb], [od]) == 0
    assert lib.cmd_exec(cmd, hint, 0, [ta, dev_q, tb], [op]) == 0
    assert np.abs(od.numpy().astype(np.float64)).max() > 0
    assert (od.numpy().view(_NP[datatype]) == op.numpy().view(_NP[datatype])).all()


def test_attention_with_a_palettized_head_projection_equals_the_dense_command(backend):
    """SCALED_DOT_PRODUCT_ATTENTION_FORWARD with the head-unifying projection's weights palettized (ccv_nnc_scaled_dot_product_attention_flash_attn.cu:123)."""
    lib = backend
    rng = np.random.default_rng(8)
    B, R, Cn, H, D, Dv = 2, 19, 23, 4, 16, 8
    q, k, v = levels(rng, (B, R, H, D), 4, np.float32), levels(rng, (B, Cn, H, D), 4, np.float32), levels(rng, (B, Cn, H, Dv), 4, np.float32)
    w, bias = levels(rng, (H * Dv, H * Dv), 5, np.float32), levels(rng, (H * Dv,), 4, np.float32)
    cmd = nnc.generic_cmd("SCALED_DOT_PRODUCT_ATTENTION_FORWARD")
    cmd.info.f1.v = 0.25                 # scaled_dot_product_attention.scale (the first field of the parameter union; tests/test_attention.py sdpa_cmd)
    cmd.info.blas.transpose_a[1] = 0     # .is_causal
    tq, tk, tv, tw, tb = make_tensors(lib, nnc.GPU_MEMORY, [q, k, v, w, bias])
    dev_q = nnc.PalettizedTensor(lib, nnc.tensor_palettize(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, H * Dv, H * Dv), 5, 256), lossless_stream(w, 5, 256, nnc.CCV_32F))
    shapes = [np.zeros((B, R, H * Dv), np.float32), np.zeros((B, H, R), np.float32), np.zeros((B, R, H, Dv), np.float32)]
    od, op = make_tensors(lib, nnc.GPU_MEMORY, shapes), make_tensors(lib, nnc.GPU_MEMORY, shapes)
    assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, [tq, tk, tv, None, tw, tb], od) == 0
    assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, [tq, tk, tv, None, dev_q, tb], op) == 0
    assert np.abs(od[0].numpy()).max() > 0
    for a, b in zip(od, op):
        assert (a.numpy().view(np.uint32) == b.numpy().view(np.uint32)).all()


def test_data_transfer_refuses_a_palettized_tensor_without_a_block_size(backend):
    lib = backend
    stream = np.zeros(256, np.uint8)
    bad = nnc.tensor_palettize(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, 8, 8), 4, 0)  # reserved = 0: no elements per block
    a, b = nnc.PalettizedTensor(lib, bad, stream), nnc.PalettizedTensor(lib, bad, stream)
    assert lib.cmd_exec(nnc.CMD_DATA_TRANSFER_FORWARD(), nnc.NO_HINT, 0, [a], [b]) == nnc.EXEC_INVALID


def test_palettized_rows_are_listed_like_the_reference_lists_them(backend):
    """CCV_QX in tensor_datatypes of exactly the rows whose reference counterparts carry it (ccv_nnc_gemm_gpu_cublas.cu, ccv_nnc_conv_gpu_cudnn.cu:482,494,
    ccv_nnc_conv_transpose_gpu_cudnn.cu:185, ccv_nnc_util_gpu_ref.cu:67,76, ccv_nnc_scaled_dot_product_attention_flash_attn.cu:459,470): the host's backend lookup matches a command's datatypes against this mask."""
    want = {"CCV_NNC_GEMM_FORWARD", "CCV_NNC_GEMM_BACKWARD", "CCV_NNC_CONVOLUTION_FORWARD", "CCV_NNC_CONVOLUTION_BACKWARD", "CCV_NNC_CONVOLUTION_TRANSPOSE_FORWARD",
            "CCV_NNC_DATA_TRANSFER_FORWARD", "CCV_NNC_DATA_TRANSFER_BACKWARD", "CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_FORWARD", "CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_BACKWARD"}
    have = {name.split("/")[0] for name, _, _, r in backend.registry() if r.tensor_datatypes & nnc.CCV_QX}
    assert have == want
