"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image transform on the hot path's input side.

The reference applies the callable returned by ``clip.load`` per sample (/root/reference/src/clip/datasets/clip_dataset.py
:110-125): Resize(n, bicubic) -> CenterCrop(n) -> RGB -> ToTensor -> Normalize(mean, std) (constants: SURVEY.md section 8 row
a16).  The arithmetic of the resize lives in a third-party dependency that is not part of /root/reference: Pillow
(``Image.resize(..., BICUBIC)``; libImaging/Resample.c, pinned here by the image's Pillow 12.2.0).  This file restates
that published algorithm -- antialiased separable filter with support 2 * max(scale, 1), cubic a = -0.5, coefficients
normalised in double and rounded to 22-bit fixed point, horizontal pass first, 8-bit clipped intermediates -- and is
pinned against Pillow itself in tests/test_preprocess.py (bit-exact on every case).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


def _bicubic(x: float, a: float = -0.5) -> float:
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coefficients(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: (int64 [out, ksize] fixed-point taps, [out, 2] (first, count))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww, ss = 0.0, 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [0.0] * ksize
        for x in range(xmax):
            k[x] = _bicubic((x + xmin - center + 0.5) * ss)
            ww += k[x]
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + k[x] * (1 << PRECISION_BITS)) if k[x] < 0 else int(0.5 + k[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def resize_bicubic_u8(arr: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """uint8 [h, w, c] -> uint8 [nh, nw, c], Pillow's two-pass resample (a pass is skipped when the size does not change)."""
    h, w, _ = arr.shape
    a = arr.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if nw != w:
        kk, b = coefficients(w, nw)
        out = np.zeros((h, nw, a.shape[2]), dtype=np.int64)
        for xx in range(nw):
            x0, cnt = b[xx]
            out[:, xx, :] = np.clip((half + (a[:, x0:x0 + cnt, :] * kk[xx, :cnt][None, :, None]).sum(1)) >> PRECISION_BITS, 0, 255)
        a = out
    if nh != h:
        kk, b = coefficients(h, nh)
        out = np.zeros((nh, a.shape[1], a.shape[2]), dtype=np.int64)
        for yy in range(nh):
            y0, cnt = b[yy]
            out[yy] = np.clip((half + (a[y0:y0 + cnt] * kk[yy, :cnt][:, None, None]).sum(0)) >> PRECISION_BITS, 0, 255)
        a = out
    return a.astype(np.uint8)


def resized_size(h: int, w: int, n: int):
    return (n, int(n * h / w)) if w <= h else (int(n * w / h), n)          # (nw, nh)


def clip_preprocess(arr: np.ndarray, n: int = 224) -> np.ndarray:
    """uint8 RGB [h, w, 3] -> float32 [3, n, n], the whole transform."""
    h, w, _ = arr.shape
    nw, nh = resized_size(h, w, n)
    r = resize_bicubic_u8(arr, nw, nh)
    left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
    x = r[top:top + n, left:left + n].astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (x - CLIP_MEAN[:, None, None]) / CLIP_STD[:, None, None]
