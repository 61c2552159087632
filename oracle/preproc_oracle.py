"""TEST INFRASTRUCTURE ONLY -- numpy / pure-Python restatement of the reference's classic pre-process loops, following
their ROW-SEQUENTIAL structure literally (the product restates them as tap tables; this file is the independent check).

  resample_area_8u   lib/ccv_resample.c:11-133    fixed-point area down-sampling, 8u -> 8u
  filter_direct_8u   lib/ccv_numeric.c:960-1034   direct correlation, replicated border, 2^14 fixed point

Pinned against the reference's own build (oracle/_ref/libccv_classic.so) by tests/test_preproc.py.  Loops are Python:
use small images."""
import numpy as np


def resample_area_8u(a, b_rows, b_cols, rows_scale, cols_scale):
    a_rows, a_cols, ch = a.shape
    scale_x, scale_y = 1.0 / cols_scale, 1.0 / rows_scale
    inv_scale_256 = int(scale_x * scale_y * 0x10000)
    xofs = []
    for dx in range(b_cols):
        fsx1 = dx * scale_x
        fsx2 = fsx1 + scale_x
        sx1, sx2 = int(fsx1 + 1.0 - 1e-6), int(fsx2)
        if sx1 > fsx1:
            xofs.append((dx, min(sx1 - 1, a_cols - 1), int((sx1 - fsx1) * 0x100)))
        for sx in range(sx1, sx2):
            xofs.append((dx, min(sx, a_cols - 1), 256))
        if fsx2 - sx2 > 1e-3:
            xofs.append((dx, min(sx2, a_cols - 1), int((fsx2 - sx2) * 256)))
    b = np.zeros((b_rows, b_cols, ch), np.uint8)
    buf = np.zeros((b_cols, ch), np.int64)
    acc = np.zeros((b_cols, ch), np.int64)
    M = 1 << 32
    dy, dy_weight_256 = 0, 0
    for sy in range(a_rows):
        for dx, sx, alpha in xofs:
            buf[dx] = (buf[dx] + a[sy, sx].astype(np.int64) * alpha) % M
        if (dy + 1) * scale_y <= sy + 1:
            beta = int(max(sy + 1 - (dy + 1) * scale_y, 0.0) * 256)
            beta1 = 256 - beta
            if sy == a_rows - 1:
                beta = int(scale_y * 256)
            else:
                dy_weight_256 = beta
            if beta <= 0:
                b[dy] = np.minimum(((acc + buf * 256) % M) // inv_scale_256, 255)
                acc[:] = 0
            else:
                b[dy] = np.minimum(((acc + buf * beta1) % M) // inv_scale_256, 255)
                acc = (buf * beta) % M
            buf[:] = 0
            dy += 1
        else:
            if sy == a_rows - 1:
                dy_weight_256 = int(scale_y * 256) - dy_weight_256
                acc = (acc + buf * dy_weight_256) % M
            else:
                dy_weight_256 += 256
                acc = (acc + buf * 256) % M
            buf[:] = 0
    while dy < b_rows:
        b[dy] = np.minimum(acc // inv_scale_256, 255)
        dy += 1
    return b


def filter_direct_8u(a, k):
    rows, cols = a.shape
    kh, kw = k.shape
    coeff = (k.astype(np.float64) * (1 << 14) + 0.5).astype(np.int64)  # (int)(v * scale + 0.5): truncation of a positive value
    coeff = np.where(k.astype(np.float64) * (1 << 14) + 0.5 < 0, -((-(k.astype(np.float64) * (1 << 14) + 0.5)).astype(np.int64)), coeff)
    yy = np.clip(np.arange(rows + kh // 2 * 2) - kh // 2, 0, rows - 1)
    xx = np.clip(np.arange(cols + kw // 2 * 2) - kw // 2, 0, cols - 1)
    pa = a[yy][:, xx].astype(np.int64)
    d = np.zeros((rows, cols), np.int64)
    for i in range(kh):
        for j in range(kw):
            d += pa[i:i + rows, j:j + cols] * coeff[i, j]
    return np.clip(d >> 14, 0, 255).astype(np.uint8)
