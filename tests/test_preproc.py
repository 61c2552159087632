"""ccv_resample / ccv_filter batch kernels against
  * the reference's OWN implementation (oracle/_ref/libccv_classic.so, built from where the sources lie), and
  * the independent numpy restatement oracle/preproc_oracle.py (which is itself pinned against the reference here).
8u area resample, the integer (8u -> 8u) bicubic and the direct 8u filter are bit-exact; float paths within 1e-5 (float -> 8u stores
within one level: the truncating store of a float sum whose last bit depends on FMA contraction); the float filter equals the
reference's FFT path wherever that is a linear convolution (whole image for one-tile images, interior otherwise)."""
import ctypes as C
import os
import sys
import numpy as np
import pytest
from ccv_amd import nnc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import preproc_oracle as PO  # noqa: E402

CCV_8U, CCV_32F = 0x1000, 0x4000
AREA, CUBIC = 0x01, 0x04


class ImageBatch(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("channels", C.c_int), ("datatype", C.c_int), ("step", C.c_long), ("image_stride", C.c_long)]


@pytest.fixture(scope="module")
def classic():
    p = os.path.join(ROOT, "oracle", "_ref", "libccv_classic.so")
    if not os.path.exists(p):
        if os.path.isdir("/root/reference/lib"):
            import subprocess
            subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref_classic.sh")])
        else:
            pytest.skip("libccv_classic.so not built")
    R = C.CDLL(p)
    R.ccv_dense_matrix_new.restype = C.c_void_p
    R.ccv_dense_matrix_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
    R.ccv_resample.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_double, C.c_int]
    R.ccv_filter.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    R.ccv_matrix_free.argtypes = [C.c_void_p]
    return R


def _mat(R, arr):
    """numpy [rows][cols][ch] -> reference ccv_dense_matrix_t* (rows padded to its step)"""
    rows, cols, ch = arr.shape
    t = (CCV_8U if arr.dtype == np.uint8 else CCV_32F) | ch
    m = R.ccv_dense_matrix_new(rows, cols, t, None, 0)
    ts = C.cast(m, C.POINTER(nnc.TensorStruct)).contents
    step = ts.info.dim[4]
    v = np.frombuffer((C.c_ubyte * (step * rows)).from_address(ts.data), dtype=np.uint8).reshape(rows, step)
    v[:, :cols * ch * arr.itemsize] = arr.reshape(rows, -1).view(np.uint8)
    return m


def _read(m, dtype, ch):
    ts = C.cast(m, C.POINTER(nnc.TensorStruct)).contents
    rows, cols, step = ts.info.dim[0], ts.info.dim[1], ts.info.dim[4]
    v = np.frombuffer((C.c_ubyte * (step * rows)).from_address(ts.data), dtype=np.uint8).reshape(rows, step)
    return v[:, :cols * ch * np.dtype(dtype).itemsize].copy().view(dtype).reshape(rows, cols, ch)


def ref_resample(R, a, out_dtype, rs, cs, kind):
    ma = _mat(R, a)
    d = C.c_void_p(0)
    R.ccv_resample(ma, C.byref(d), (CCV_8U if out_dtype == np.uint8 else CCV_32F), rs, cs, kind)
    out = _read(d, out_dtype, a.shape[2])
    R.ccv_matrix_free(ma)
    R.ccv_matrix_free(d)
    return out


def dev_batch(L, arrs):
    """Upload a list of equally-shaped images as one batch (rows padded to 4 bytes like the reference's rasters)."""
    rows, cols, ch = arrs[0].shape
    step = (cols * ch * arrs[0].itemsize + 3) & ~3
    host = np.zeros((len(arrs), rows, step), np.uint8)
    for i, a in enumerate(arrs):
        host[i, :, :cols * ch * a.itemsize] = a.reshape(rows, -1).view(np.uint8)
    n = host.nbytes
    p = L.malloc(0, (n + 127) & ~127)
    L.memcpy(p, nnc.GPU_MEMORY, host.ctypes.data, nnc.CPU_MEMORY, n)
    desc = ImageBatch(rows, cols, ch, CCV_8U if arrs[0].dtype == np.uint8 else CCV_32F, step, step * rows)
    return p, desc, host.shape


def dev_read(L, p, desc, count, dtype):
    step, rows, cols, ch = desc.step, desc.rows, desc.cols, desc.channels
    host = np.zeros((count, rows, step), np.uint8)
    L.memcpy(host.ctypes.data, nnc.CPU_MEMORY, p, nnc.GPU_MEMORY, host.nbytes)
    return [host[i, :, :cols * ch * np.dtype(dtype).itemsize].copy().view(dtype).reshape(rows, cols, ch) for i in range(count)]


def our_resample(L, arrs, out_shape, out_dtype, rs, cs, kind):
    L.dll.nnc_mi355x_resample_batch.argtypes = [C.c_void_p, ImageBatch, C.c_void_p, ImageBatch, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
    pa, da, _ = dev_batch(L, arrs)
    zeros = [np.zeros(out_shape + (arrs[0].shape[2],), out_dtype) for _ in arrs]
    pb, db, _ = dev_batch(L, zeros)
    st = L.stream_new(0)
    r = L.dll.nnc_mi355x_resample_batch(pa, da, pb, db, len(arrs), rs, cs, kind, st)
    assert r == 0, r
    L.stream_wait(st)
    out = dev_read(L, pb, db, len(arrs), out_dtype)
    L.stream_free(st)
    L.free(0, pa)
    L.free(0, pb)
    return out


RESAMPLE_CASES = [
    # (rows, cols, ch) -> (out rows, out cols)
    ((37, 41, 3), (18, 20)), ((64, 48, 3), (32, 24)), ((50, 50, 1), (17, 23)), ((33, 47, 4), (30, 11)), ((96, 96, 3), (45, 45)),
]


def _scales(a_shape, o):
    return o[0] / a_shape[0], o[1] / a_shape[1]


@pytest.mark.parametrize("shape,out", RESAMPLE_CASES)
def test_numpy_oracle_area_8u_matches_reference(classic, shape, out):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    rs, cs = _scales(shape, out)
    np.testing.assert_array_equal(PO.resample_area_8u(a, out[0], out[1], rs, cs), ref_resample(classic, a, np.uint8, rs, cs, AREA))


@pytest.mark.parametrize("shape,out", RESAMPLE_CASES)
def test_resample_area_8u_bit_exact(backend, classic, shape, out):
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(3)]
    rs, cs = _scales(shape, out)
    got = our_resample(backend, imgs, out, np.uint8, rs, cs, AREA)
    for g, a in zip(got, imgs):
        np.testing.assert_array_equal(g, ref_resample(classic, a, np.uint8, rs, cs, AREA))
        np.testing.assert_array_equal(g, PO.resample_area_8u(a, out[0], out[1], rs, cs))


@pytest.mark.parametrize("shape,out,fill", [((64, 48, 3), (9, 7), 255), ((64, 48, 3), (32, 24), 255), ((61, 52, 1), (4, 5), None), ((128, 64, 4), (8, 4), None), ((80, 80, 3), (79, 77), None), ((48, 64, 3), (47, 3), 0)])
def test_resample_area_8u_extremes_bit_exact(backend, classic, shape, out, fill):
    """The row-per-workgroup kernel (round 5) at its edges: saturated and empty images (the quotient estimate and its correction at 255 and 0), sixteen-fold and
    barely-any reduction (many taps / almost no taps per output), 16-byte and 4-byte row pitches, 1 / 3 / 4 channels -- bit-exact against ccv_resample."""
    rng = np.random.default_rng(41)
    imgs = [np.full(shape, fill, np.uint8) if fill is not None else rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(2)]
    if fill is None:
        imgs[1][::2] = 255  # rows of saturation next to noise
    rs, cs = _scales(shape, out)
    got = our_resample(backend, imgs, out, np.uint8, rs, cs, AREA)
    for g, a in zip(got, imgs):
        np.testing.assert_array_equal(g, ref_resample(classic, a, np.uint8, rs, cs, AREA))


@pytest.mark.gpu
def test_resample_area_8u_imagenet_size_bit_exact(gpu_lib):
    """256 x 480 x 480 x 3 -> 224 x 224 (the size the bench quotes) on the MI355X, every image against the numpy restatement of ccv_resample's 8-bit area path
    (oracle/preproc_oracle.py, pinned to libccv_classic.so by the tests above), and the old one-lane-per-byte kernel against the new one."""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (480, 480, 3), dtype=np.uint8) for _ in range(8)]
    rs = cs = 224 / 480
    got = our_resample(gpu_lib, imgs, (224, 224), np.uint8, rs, cs, AREA)
    for g, a in zip(got[:3], imgs[:3]):
        np.testing.assert_array_equal(g, PO.resample_area_8u(a, 224, 224, rs, cs))
    code = "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_preproc as T; from ccv_amd import nnc; rng = np.random.default_rng(5); imgs = [rng.integers(0, 256, (480, 480, 3), dtype=np.uint8) for _ in range(8)]; got = T.our_resample(nnc.load(), imgs, (224, 224), np.uint8, 224 / 480, 224 / 480, T.AREA); np.save(sys.argv[1], np.stack(got))" % (ROOT, os.path.join(ROOT, "tests"))
    path = os.path.join(ROOT, "gpurun_out", "resample_old_kernel.npy")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, NNC_MI355X_RESAMPLE_ROWS="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    np.testing.assert_array_equal(np.load(path), np.stack(got))


@pytest.mark.parametrize("shape,out", RESAMPLE_CASES[:3])
@pytest.mark.parametrize("src,dst", [(np.uint8, np.float32), (np.float32, np.float32), (np.float32, np.uint8)])
def test_resample_area_float(backend, classic, shape, out, src, dst):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, shape, dtype=np.uint8) if src == np.uint8 else (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
    rs, cs = _scales(shape, out)
    got = our_resample(backend, [a, a], out, dst, rs, cs, AREA)[1]
    want = ref_resample(classic, a, dst, rs, cs, AREA)
    if dst == np.uint8:
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1  # truncating store of a float sum: last-bit ties
    else:
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("shape,out", [((20, 24, 3), (45, 50)), ((31, 17, 1), (64, 40)), ((16, 16, 4), (33, 35))])
@pytest.mark.parametrize("src,dst", [(np.uint8, np.float32), (np.float32, np.float32), (np.uint8, np.uint8)])
def test_resample_cubic(backend, classic, shape, out, src, dst):
    rng = np.random.default_rng(6)
    a = rng.integers(0, 256, shape, dtype=np.uint8) if src == np.uint8 else (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
    rs, cs = _scales(shape, out)
    got = our_resample(backend, [a], out, dst, rs, cs, CUBIC)[0]
    want = ref_resample(classic, a, dst, rs, cs, CUBIC)
    if dst == np.uint8 and src == np.uint8:
        np.testing.assert_array_equal(got, want)  # the integer-only bicubic (6-bit coefficients, ccv_resample.c:343-431): byte work, bit-exact
    elif dst == np.uint8:
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    else:
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-3)


def our_filter(L, arrs, k):
    L.dll.nnc_mi355x_filter_batch.argtypes = [C.c_void_p, ImageBatch, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, ImageBatch, C.c_int, C.c_void_p]
    pa, da, _ = dev_batch(L, arrs)
    pd, dd, _ = dev_batch(L, [np.zeros_like(x) for x in arrs])
    kk = np.ascontiguousarray(k, dtype=np.float32)
    kch = 1 if kk.ndim == 2 else kk.shape[2]
    st = L.stream_new(0)
    r = L.dll.nnc_mi355x_filter_batch(pa, da, kk.ctypes.data, kk.shape[0], kk.shape[1], kch, pd, dd, len(arrs), st)
    assert r == 0, r
    L.stream_wait(st)
    out = dev_read(L, pd, dd, len(arrs), arrs[0].dtype)
    L.stream_free(st)
    L.free(0, pa)
    L.free(0, pd)
    return out


def ref_filter(R, a, k):
    ma, mk = _mat(R, a), _mat(R, k.reshape(k.shape[0], k.shape[1], 1).astype(np.float32))
    d = C.c_void_p(0)
    R.ccv_filter(ma, mk, C.byref(d), 0, 0)
    out = _read(d, a.dtype, a.shape[2])
    for m in (ma, mk, d):
        R.ccv_matrix_free(m)
    return out


@pytest.mark.parametrize("shape,ksize", [((24, 31), (3, 3)), ((40, 40), (5, 3)), ((19, 23), (3, 5))])
def test_filter_8u_bit_exact(backend, classic, shape, ksize):
    rng = np.random.default_rng(8)
    imgs = [rng.integers(0, 256, shape + (1,), dtype=np.uint8) for _ in range(2)]
    k = rng.random(ksize, dtype=np.float32)
    k /= k.sum()
    got = our_filter(backend, imgs, k)
    for g, a in zip(got, imgs):
        np.testing.assert_array_equal(g, ref_filter(classic, a, k))
        np.testing.assert_array_equal(g[:, :, 0], PO.filter_direct_8u(a[:, :, 0], k))


@pytest.mark.parametrize("shape,ksize", [((30, 28), (3, 3)), ((44, 40), (11, 11)), ((25, 33), (5, 7))])
def test_filter_f32_interior(backend, classic, shape, ksize):
    rng = np.random.default_rng(9)
    a = rng.random(shape + (1,), dtype=np.float32)
    k = rng.random(ksize, dtype=np.float32)
    got = our_filter(backend, [a], k)[0]
    want = ref_filter(classic, a, k)  # tiled-FFT path: at the borders of an image larger than one tile its rows wrap around
    kh, kw = ksize
    np.testing.assert_allclose(got[kh:-kh, kw:-kw], want[kh:-kh, kw:-kw], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,ksize", [((30, 28), (3, 3)), ((44, 40), (11, 11)), ((25, 33), (5, 7)), ((64, 50), (4, 6)), ((37, 90), (7, 3)), ((100, 100), (9, 9)), ((23, 61), (8, 5)), ((19, 19), (2, 2))])
def test_filter_f32_whole_image_with_the_reference_tiling(backend, classic, shape, ksize):
    """Round 5: images that need SEVERAL tiles of the reference's FFT path.  At their borders the reference's circular
    convolution wraps the far rows / columns of a tile's window in (lib/ccv_numeric.c:846-925); the kernel now reproduces that tiling (window origin, extent and
    read position per output row / column, filter_axis_map in img_preproc.cpp), so EVERY output element is compared, borders included -- odd and even windows,
    one and several tiles per axis, the clamped last tile and its edge block."""
    rng = np.random.default_rng(12)
    a = rng.random(shape + (1,), dtype=np.float32)
    k = rng.random(ksize, dtype=np.float32)
    got = our_filter(backend, [a], k)[0]
    want = ref_filter(classic, a, k)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-4 * float(k.sum()))


@pytest.mark.parametrize("shape,ksize", [((12, 14), (7, 7)), ((10, 10), (10, 10)), ((11, 11), (11, 11)), ((9, 13), (6, 7)), ((16, 12), (9, 8))])
def test_filter_f32_whole_image_where_the_reference_fft_is_linear(backend, classic, shape, ksize):
    """Images that fit ONE tile of the reference's FFT path with room for the kernel (image + kernel - 1 <= tile:
    ccv_numeric.c:775-776): the circular convolution is then the linear one and EVERY output element is defined -- zero border,
    centre tap (size - 1) / 2 for even and odd windows.  The whole image is compared, borders included."""
    rng = np.random.default_rng(10)
    a = rng.random(shape + (1,), dtype=np.float32)
    k = rng.random(ksize, dtype=np.float32)
    got = our_filter(backend, [a], k)[0]
    want = ref_filter(classic, a, k)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * float(np.abs(want).max()))


@pytest.mark.parametrize("n", [10, 11])
def test_filter_centre_point_of_the_reference_unit_tests(backend, n):
    """test/unit/numeric.tests.c:112-150 replayed: x = 0 .. n^2 - 1, y = x reversed, ccv_filter(x, y)[centre] = sum_i (n^2 - 1 - i) * i
    at the centre (n - 1) / 2 -- for the even window too ("hint: (size - 1) / 2"), tolerance 0.1 as there."""
    x = np.arange(n * n, dtype=np.float32).reshape(n, n, 1)
    y = x[::-1, ::-1, 0].copy()
    d = our_filter(backend, [x], y)[0]
    want = float(sum((n * n - 1 - i) * i for i in range(n * n)))
    c = (n - 1) // 2
    assert abs(float(d[c, c, 0]) - want) <= 0.1 * max(1.0, want * 1e-6), (d[c, c, 0], want)
