"""GPU data pipeline, first slice: the pixel half of the reference's random jitter
(lib/nnc/ccv_cnnp_dataframe_addons.c:265-366) + one-hot labels (:378-) as batch kernels.

The DECISIONS stay on the host, as integer arithmetic on the generator's draws (restated in `plan()` below from :276-330 -- the
C-ABI takes them as input, so a maintainer's binding keeps the reference's own sfmt stream); the PIXELS are checked against the
reference's own classic functions run image by image (oracle/_ref/libccv_classic.so: ccv_slice -> ccv_resample(CCV_32F, area |
cubic) -> ccv_flip -> normalise -> late ccv_slice), then laid out as the trainer's batch tensor (NHWC / NCHW, fp32 / half)."""
import ctypes as C
import os
import numpy as np
import pytest
from ccv_amd import nnc
from test_preproc import classic, _mat, _read, CCV_8U, CCV_32F  # noqa: F401  (fixture + helpers)

F = np.float32
AREA, CUBIC = 0x01, 0x04


class ColorOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("v", C.c_float * 3)]


class JitterImage(C.Structure):
    _fields_ = [("offset", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int), ("step", C.c_int),
                ("slice_x", C.c_int), ("slice_y", C.c_int), ("slice_rows", C.c_int), ("slice_cols", C.c_int),
                ("resize_rows", C.c_int), ("resize_cols", C.c_int), ("crop_x", C.c_int), ("crop_y", C.c_int), ("flip", C.c_int),
                ("color_ops", C.c_int), ("color", ColorOp * 4)]


BRIGHTNESS, SATURATION, CONTRAST, LIGHTING = 1, 2, 3, 4


class JitterParams(C.Structure):
    _fields_ = [("out_rows", C.c_int), ("out_cols", C.c_int), ("channels", C.c_int), ("mean", C.c_float * 3), ("inv_std", C.c_float * 3),
                ("format", C.c_int), ("datatype", C.c_int)]


def clamp(v, lo, hi):
    return max(lo, min(hi, v))


def plan(rng, rows, cols, rmin, rmax, size, aspect=0.0, symmetric=True, center_crop=False):
    """The reference's decisions for one image (:276-330), u = uniform [0, 1] draws in the reference's order."""
    u = lambda: float(rng.random())
    resize = clamp(int(u() * (rmax - rmin) + 0.5) + rmin, rmin, rmax)
    rr = max(resize, int(rows * F(resize) / cols + 0.5))
    rc = max(resize, int(cols * F(resize) / rows + 0.5))
    if aspect > 0:
        ar = float(np.sqrt(np.exp((u() * 2 - 1) * np.log(aspect))))
        rr, rc = int(rr * ar + 0.5), int(rc / ar + 0.5)
    srows, scols = size
    need_crop = rc != scols or rr != srows
    d = dict(slice=(0, 0, rows, cols), resize=(rr, rc), crop=(0, 0))
    if need_crop:
        cx = (rc - scols + 1) // 2 if center_crop else int(u() * (rc - scols + 1))
        cx = clamp(cx, min(0, rc - scols), max(0, rc - scols))
        cy = (rr - srows + 1) // 2 if center_crop else int(u() * (rr - srows + 1))
        cy = clamp(cy, min(0, rr - srows), max(0, rr - srows))
        if rc >= scols and rr >= srows:  # crop first, then scale (:316-326)
            sx, sy = F(cols) / rc, F(rows) / rr
            sc, sr = int(scols * sx + 0.5), int(srows * sy + 0.5)
            x = clamp(int(cx * sx + 0.5), 0, cols - sc)
            y = clamp(int(cy * sy + 0.5), 0, rows - sr)
            d = dict(slice=(x, y, sr, sc), resize=(srows, scols), crop=(0, 0))
        else:
            d["crop"] = (cx, cy)
    d["flip"] = bool(symmetric and (int(rng.integers(0, 2)) == 0))
    return d


def ref_color(R, d, ops):
    """_ccv_cnnp_image_manip (ccv_cnnp_dataframe_addons.c:213-253) on the float matrix handle d, in place, with the reference's own
    ccv_scale / ccv_saturation / ccv_contrast; lighting restated from :187-198 (a static function there: float add, clamp to [0, 255])."""
    for kind, v in ops:
        if kind == BRIGHTNESS:
            R.ccv_scale(d, C.byref(d), 0, C.c_double(float(F(v[0]))))
        elif kind == SATURATION:
            R.ccv_saturation(d, C.byref(d), 0, C.c_double(float(F(v[0]))))
        elif kind == CONTRAST:
            R.ccv_contrast(d, C.byref(d), 0, C.c_double(float(F(v[0]))))
        else:
            ts = C.cast(d, C.POINTER(nnc.TensorStruct)).contents   # ccv_dense_matrix_t shares the header layout (rows, cols, ..., step)
            rows, cols, step = ts.info.dim[0], ts.info.dim[1], ts.info.dim[4]
            raw = np.frombuffer((C.c_ubyte * (step * rows)).from_address(ts.data), dtype=np.uint8).reshape(rows, step)
            a = raw[:, :cols * 12].view(F).reshape(rows, cols, 3)
            a[:] = np.clip(a + np.asarray(v, F), F(0), F(255))


def ref_pipeline(R, img, p, size, mean, inv_std, ops=()):
    x, y, sr, sc = p["slice"]
    m = _mat(R, img)
    cur = m
    if (x, y, sr, sc) != (0, 0, img.shape[0], img.shape[1]):
        s = C.c_void_p(0)
        R.ccv_slice(m, C.byref(s), 0, y, x, sr, sc)
        cur = s
    rr, rc = p["resize"]
    d = C.c_void_p(0)
    if sr >= rr and sc >= rc and (sr, sc) != (rr, rc):
        R.ccv_resample(cur, C.byref(d), CCV_32F, rr / sr, rc / sc, AREA)
    elif (sr, sc) != (rr, rc):
        R.ccv_resample(cur, C.byref(d), CCV_32F, rr / sr, rc / sc, CUBIC)
    else:
        R.ccv_shift(cur, C.byref(d), CCV_32F, 0, 0)
    if p["flip"]:
        R.ccv_flip(d, C.byref(d), 0, 0x01)
    if ops:
        ref_color(R, d, ops)
    out = _read(d, F, 3)
    for h in {m, getattr(cur, "value", None), d.value} - {None}:
        pass  # (matrices are leaked on purpose: the reference's cache owns some of them)
    out = (out - np.asarray(mean, F)) * np.asarray(inv_std, F)
    srows, scols = size
    cx, cy = p["crop"]
    win = np.zeros((srows, scols, 3), F)  # late crop: ccv_slice zero-fills what hangs over (:357-361)
    ys, xs = max(0, -cy), max(0, -cx)
    ye, xe = min(srows, out.shape[0] - cy), min(scols, out.shape[1] - cx)
    if ye > ys and xe > xs:
        win[ys:ye, xs:xe] = out[cy + ys:cy + ye, cx + xs:cx + xe]
    return win


@pytest.mark.parametrize("fmt,dtype", [("NHWC", "f32"), ("NCHW", "f32"), ("NCHW", "f16")])
def test_jitter_batch_against_the_reference_functions(backend, classic, fmt, dtype):
    L, R = backend, classic
    R.ccv_slice.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    R.ccv_flip.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    R.ccv_shift.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(31)
    size = (24, 24)
    shapes = [(40, 52), (61, 37), (24, 24), (20, 30), (33, 33), (18, 64), (90, 70), (25, 26)]
    # resize ranges chosen so that every branch occurs: shrink (area), enlarge (cubic), same size, crop-first and late crop with overhang
    ranges = [(26, 32), (26, 32), (24, 24), (24, 28), (16, 20), (28, 30), (26, 40), (24, 24)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]
    plans = [plan(rng, s[0], s[1], r[0], r[1], size, aspect=(0.5 if i % 3 == 0 else 0.0), center_crop=(i == 4)) for i, (s, r) in enumerate(zip(shapes, ranges))]
    mean, std = (123.68, 116.779, 103.939), (58.393, 57.12, 57.375)
    inv_std = tuple(1.0 / s for s in std)
    want = np.stack([ref_pipeline(R, a, p, size, mean, inv_std) for a, p in zip(imgs, plans)])
    kinds = set()
    for a, p in zip(imgs, plans):
        _, _, sr, sc = p["slice"]
        rr, rc = p["resize"]
        kinds.add("area" if sr >= rr and sc >= rc and (sr, sc) != (rr, rc) else "cubic" if (sr, sc) != (rr, rc) else "same")
        kinds.add("late-crop" if p["crop"] != (0, 0) or (rr, rc) != size else "crop-first")
    assert {"area", "cubic", "same", "late-crop", "crop-first"} <= kinds, kinds
    # one source buffer, images back to back with 4-byte row pitch
    descs = (JitterImage * len(imgs))()
    blobs, off = [], 0
    for i, (a, p) in enumerate(zip(imgs, plans)):
        rows, cols = a.shape[:2]
        step = (cols * 3 + 3) & ~3
        buf = np.zeros((rows, step), np.uint8)
        buf[:, :cols * 3] = a.reshape(rows, -1)
        x, y, sr, sc = p["slice"]
        descs[i] = JitterImage(off, rows, cols, step, x, y, sr, sc, p["resize"][0], p["resize"][1], p["crop"][0], p["crop"][1], int(p["flip"]))
        blobs.append(buf.reshape(-1))
        off += (buf.size + 15) & ~15
        blobs.append(np.zeros(off - sum(b.size for b in blobs), np.uint8))
    host = np.concatenate(blobs)
    src = L.malloc(0, (host.nbytes + 127) & ~127)
    L.memcpy(src, nnc.GPU_MEMORY, host.ctypes.data, nnc.CPU_MEMORY, host.nbytes)
    T = np.float16 if dtype == "f16" else F
    n = len(imgs)
    out_shape = (n, size[0], size[1], 3) if fmt == "NHWC" else (n, 3, size[0], size[1])
    out = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC if fmt == "NHWC" else nnc.NCHW, nnc.CCV_16F if dtype == "f16" else nnc.CCV_32F, out_shape, 0), np.full(out_shape, 7, T))
    params = JitterParams(size[0], size[1], 3, (C.c_float * 3)(*mean), (C.c_float * 3)(*inv_std), nnc.NHWC if fmt == "NHWC" else nnc.NCHW, nnc.CCV_16F if dtype == "f16" else nnc.CCV_32F)
    L.dll.nnc_mi355x_jitter_batch.argtypes = [C.c_void_p, C.POINTER(JitterImage), C.c_int, JitterParams, C.c_void_p, C.c_void_p]
    st = L.stream_new(0)
    assert L.dll.nnc_mi355x_jitter_batch(src, descs, n, params, out.ptr, st) == 0
    L.stream_wait(st)
    got = out.numpy().astype(F)
    if fmt == "NCHW":
        got = got.transpose(0, 2, 3, 1)
    if dtype == "f16":
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    # one-hot rows with the trainer's label smoothing (bin/nnc/imagenet.c:394-395)
    labels = rng.integers(0, 10, n).astype(np.int32)
    eta = 0.1
    oh = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NCHW, nnc.CCV_16F if dtype == "f16" else nnc.CCV_32F, (n, 10), 0), np.zeros((n, 10), T))
    L.dll.nnc_mi355x_one_hot_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    assert L.dll.nnc_mi355x_one_hot_batch(labels.ctypes.data, n, 10, 1 - eta + eta / 10, eta / 10, nnc.CCV_16F if dtype == "f16" else nnc.CCV_32F, oh.ptr, st) == 0
    L.stream_wait(st)
    want_oh = np.full((n, 10), eta / 10, F)
    want_oh[np.arange(n), labels] = 1 - eta + eta / 10
    np.testing.assert_array_equal(oh.numpy(), want_oh.astype(T))
    L.stream_free(st)
    L.free(0, src)


def test_jitter_colour_operations_against_the_reference_functions(backend, classic):
    """brightness / saturation / contrast / lighting in shuffled orders (what _ccv_cnnp_image_manip's sfmt_genrand_shuffle produces),
    with the factors a generator would have drawn, against ccv_scale / ccv_saturation / ccv_contrast run in that order on the
    resampled float image; contrast uses the mean of the WHOLE resampled image at that point (after a clamping lighting step in one
    of the cases)."""
    L, R = backend, classic
    R.ccv_slice.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    R.ccv_flip.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    R.ccv_shift.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    for f in (R.ccv_scale, R.ccv_saturation, R.ccv_contrast):
        f.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_double]
    rng = np.random.default_rng(41)
    size = (20, 20)
    shapes = [(40, 52), (33, 33), (18, 30), (20, 20), (64, 48)]
    ranges = [(22, 28), (20, 24), (24, 28), (20, 20), (24, 30)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]
    imgs[2][:6] = 255   # saturated rows: the lighting clamp bites
    imgs[2][6:9] = 0
    plans = [plan(rng, s[0], s[1], r[0], r[1], size) for s, r in zip(shapes, ranges)]
    colour = [
        [(CONTRAST, (1.3, 0, 0)), (BRIGHTNESS, (0.9, 0, 0)), (SATURATION, (1.2, 0, 0)), (LIGHTING, (3.1, -2.2, 1.4))],
        [(SATURATION, (0.8, 0, 0)), (CONTRAST, (0.75, 0, 0))],
        [(LIGHTING, (-6.0, 4.5, 7.0)), (CONTRAST, (1.25, 0, 0)), (BRIGHTNESS, (1.1, 0, 0))],   # contrast about the mean of the CLAMPED image
        [],
        [(BRIGHTNESS, (1.2, 0, 0))],
    ]
    mean, std = (123.68, 116.779, 103.939), (58.393, 57.12, 57.375)
    inv_std = tuple(1.0 / s for s in std)
    want = np.stack([ref_pipeline(R, a, p, size, mean, inv_std, ops) for a, p, ops in zip(imgs, plans, colour)])
    descs = (JitterImage * len(imgs))()
    blobs, off = [], 0
    for i, (a, p, ops) in enumerate(zip(imgs, plans, colour)):
        rows, cols = a.shape[:2]
        step = (cols * 3 + 3) & ~3
        buf = np.zeros((rows, step), np.uint8)
        buf[:, :cols * 3] = a.reshape(rows, -1)
        x, y, sr, sc = p["slice"]
        d = JitterImage(off, rows, cols, step, x, y, sr, sc, p["resize"][0], p["resize"][1], p["crop"][0], p["crop"][1], int(p["flip"]))
        d.color_ops = len(ops)
        for k, (kind, v) in enumerate(ops):
            d.color[k].kind = kind
            for e in range(3):
                d.color[k].v[e] = v[e]
        descs[i] = d
        blobs.append(buf.reshape(-1))
        off += (buf.size + 15) & ~15
        blobs.append(np.zeros(off - sum(b.size for b in blobs), np.uint8))
    host = np.concatenate(blobs)
    src = L.malloc(0, (host.nbytes + 127) & ~127)
    L.memcpy(src, nnc.GPU_MEMORY, host.ctypes.data, nnc.CPU_MEMORY, host.nbytes)
    n = len(imgs)
    out = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, (n, size[0], size[1], 3), 0), np.zeros((n, size[0], size[1], 3), F))
    params = JitterParams(size[0], size[1], 3, (C.c_float * 3)(*mean), (C.c_float * 3)(*inv_std), nnc.NHWC, nnc.CCV_32F)
    L.dll.nnc_mi355x_jitter_batch.argtypes = [C.c_void_p, C.POINTER(JitterImage), C.c_int, JitterParams, C.c_void_p, C.c_void_p]
    st = L.stream_new(0)
    assert L.dll.nnc_mi355x_jitter_batch(src, descs, n, params, out.ptr, st) == 0
    L.stream_wait(st)
    np.testing.assert_allclose(out.numpy(), want, rtol=1e-5, atol=3e-5)
    L.stream_free(st)
    L.free(0, src)
