"""Host-side coefficient tables for the GPU Lanczos-3 resampler.

The reference resizes every pyramid level with ``PIL.Image.resize(..., resample=LANCZOS)`` on the host
(quick_start/coarseAlignFeatMatch.py:80-90).  Pillow's resampler (libImaging/Resample.c) is a separable
two-pass filter in fixed point: per output coordinate a window [xmin, xmin+xmax) of source pixels and int32
weights round(w * 2^22); each pass accumulates ``2^21 + sum(pixel * weight)`` in int32, shifts right by 22 and
clamps to uint8.  The device kernels (rfx_lanczos_pass_u8) do exactly that integer arithmetic, so the result is
bit-identical to Pillow's provided the weight tables are: they are computed here, in double precision with the
same operation order as Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` (libm ``sin`` via
``math.sin``), once per (input size, output size) and cached.
"""
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


@functools.lru_cache(maxsize=256)
def coeffs(in_size, out_size):
    """-> (bounds int32 (out_size, 2) [xmin, count], weights int32 (out_size, ksize), ksize)."""
    in0, in1 = 0.0, float(in_size)
    filterscale = scale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return bounds, kk, ksize


def plan(in_w, in_h, out_w, out_h):
    """Pass plan of ImagingResample for a full-image box: which passes run, the row window of the horizontal
    pass and the (shifted) vertical bounds.  Returns dict or None when the size is unchanged (Pillow copies)."""
    if (in_w, in_h) == (out_w, out_h):
        return None
    need_h, need_v = out_w != in_w, out_h != in_h
    bh, kh, ksh = coeffs(in_w, out_w)
    bv, kv, ksv = coeffs(in_h, out_h)
    y_first = int(bv[0, 0])
    y_last = int(bv[out_h - 1, 0] + bv[out_h - 1, 1])
    bv2 = bv.copy()
    if need_h:
        bv2[:, 0] -= y_first
    return dict(need_h=need_h, need_v=need_v, bounds_h=bh, kk_h=kh, ks_h=ksh, bounds_v=bv2, kk_v=kv, ks_v=ksv,
                y_first=y_first if need_h else 0, rows_h=(y_last - y_first) if need_h else in_h)


def apply_numpy(img, out_w, out_h):
    """Integer two-pass resample of a uint8 HWC array with the tables above (CPU check of the tables against
    Pillow itself; the product path runs the same arithmetic in rfx_lanczos_pass_u8)."""
    h, w, c = img.shape
    p = plan(w, h, out_w, out_h)
    if p is None:
        return img.copy()
    cur = img.astype(np.int64)
    if p["need_h"]:
        rows = cur[p["y_first"]:p["y_first"] + p["rows_h"]]
        out = np.empty((rows.shape[0], out_w, c), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = p["bounds_h"][xx]
            acc = (1 << (PRECISION_BITS - 1)) + (rows[:, x0:x0 + n, :] * p["kk_h"][xx, :n].astype(np.int64)[None, :, None]).sum(1)
            out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out.astype(np.int64)
    if p["need_v"]:
        out = np.empty((out_h, cur.shape[1], c), dtype=np.uint8)
        for yy in range(out_h):
            y0, n = p["bounds_v"][yy]
            acc = (1 << (PRECISION_BITS - 1)) + (cur[y0:y0 + n] * p["kk_v"][yy, :n].astype(np.int64)[:, None, None]).sum(0)
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out
    return cur.astype(np.uint8)
