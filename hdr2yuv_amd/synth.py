"""Synthetic frames of SURVEY.md 8c/8d (host-side input generation for bench/tests).

uint32 s = 12345 + frame; for plane G, B, R in turn, for i in row-major order:
s = s*1664525 + 1013904223; v = (float)(s >> 8) / 2^24; then plane[0] = 0.0 and
plane[1] = 1.0 so that (int)min = 0 and (int)max = 1.  numpy, vectorised through
the closed form of the LCG (a^i and c * sum a^j, both mod 2^32).
"""
from __future__ import annotations

import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)
_cache = {}


def _lcg_tables(n: int):
    t = _cache.get(n)
    if t is None:
        with np.errstate(over="ignore"):
            ai = np.cumprod(np.full(n, _A, dtype=np.uint32), dtype=np.uint32)          # a^(i+1)
            geo = np.cumsum(np.concatenate(([np.uint32(1)], ai[:-1])), dtype=np.uint32)  # sum_{j<=i} a^j
        t = (ai, geo)
        _cache.clear()
        _cache[n] = t
    return t


def synth_frame(width: int, height: int, frame: int = 0, f16: bool = False):
    """Three planes (G,B,R): float32, or float16 bit patterns (uint16) when f16."""
    n = width * height
    ai, geo = _lcg_tables(3 * n)
    with np.errstate(over="ignore"):
        s = ai * np.uint32(12345 + frame) + geo * _C
    v = (s >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    planes = [v[c * n:(c + 1) * n].copy() for c in range(3)]
    if f16:
        planes = [p.astype(np.float16) for p in planes]
    for p in planes:
        p[0], p[1] = 0.0, 1.0
    if f16:
        return [p.view(np.uint16) for p in planes]
    return planes
