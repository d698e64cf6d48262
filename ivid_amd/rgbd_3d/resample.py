"""Host-side coefficient tables for the device LANCZOS resolve.

aggregate_conditions (rgbd_3d/utils.py:454) shrinks the 3x-supersampled colour with
`Image.fromarray(to8b(color)).resize((S,S), LANCZOS)`: Pillow's 8-bit separable resampler.  To stay
bit-compatible the device kernel runs the same two integer passes; this module restates Pillow's
Resample.c coefficient set-up (precompute_coeffs + normalize_coeffs_8bpc: double-precision
windowed-sinc weights, normalised per output pixel, rounded to 22 fractional bits)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    return 1.0 if x == 0.0 else math.sin(x * math.pi) / (x * math.pi)


def _lanczos3(x):
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def lanczos_tables(in_size, out_size):
    """-> bounds int32 [out,2] (first source index, tap count), coeffs int32 [out,ksize], ksize."""
    scale = in_size / out_size
    fscale = scale if scale > 1.0 else 1.0
    support = 3.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / fscale
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = 0 if lo < 0 else lo
        hi = int(center + support + 0.5)
        hi = in_size if hi > in_size else hi
        n = hi - lo
        w = [_lanczos3((j + lo - center + 0.5) * inv) for j in range(n)]
        tot = sum(w)
        for j in range(n):
            v = w[j] / tot if tot != 0.0 else w[j]
            coeffs[o, j] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[o] = (lo, n)
    return bounds, coeffs, ksize


def resample8_reference(img, out_size):
    """Numpy emulation of the device kernel's arithmetic ([H,W,3] uint8 -> [out,out,3] uint8); used by the CPU
    tests to check these tables against the real Pillow."""
    def one_pass(a, axis):
        n_in = a.shape[axis]
        bounds, coeffs, _ = lanczos_tables(n_in, out_size)
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        out = np.empty((out_size,) + a.shape[1:], dtype=np.uint8)
        for o in range(out_size):
            lo, n = bounds[o]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(coeffs[o, :n].astype(np.int64), a[lo:lo + n], axes=(0, 0))
            out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        return np.moveaxis(out, 0, axis)
    return one_pass(one_pass(img, 1), 0)
