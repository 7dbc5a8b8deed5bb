"""ctypes bindings for the oracle -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke().
The product package (hdr2yuv_amd) must never import this module.

`Oracle`  -> oracle/liboracle.so      our CPU restatement (h2y_oracle.c)
`Ref`     -> oracle/_ref/libh2y_ref.so the reference's own convert.cpp/common.cpp
             object code behind ref_shim.cpp (built in the container only).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference"


class H2YDesc(C.Structure):
    """Mirror of h2y_desc, include/hdr2yuv_hip.h."""

    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("in_sample_type", C.c_int32),
        ("src_bit_depth", C.c_int32),
        ("dst_bit_depth", C.c_int32),
        ("src_transfer", C.c_int32),
        ("dst_transfer", C.c_int32),
        ("src_matrix", C.c_int32),
        ("dst_matrix", C.c_int32),
        ("src_primaries", C.c_int32),
        ("dst_primaries", C.c_int32),
        ("dst_full_range", C.c_int32),
        ("dst_chroma_format_idc", C.c_int32),
        ("chroma_resampler_type", C.c_int32),
        ("stats_override", C.c_int32),
        ("floor", C.c_int32 * 3),
        ("ceiling", C.c_int32 * 3),
    ]


SAMPLE_U16, SAMPLE_F32, SAMPLE_F16 = 1, 2, 3
CHROMA_420, CHROMA_444 = 1, 3
TRANSFER_LINEAR, TRANSFER_PQ = 8, 16
MATRIX_GBR, MATRIX_BT709, MATRIX_BT2020NC, MATRIX_YDZDX, MATRIX_Y500, MATRIX_Y100 = 0, 1, 9, 11, 12, 13


def make_desc(width, height, *, sample=SAMPLE_F32, src_depth=32, dst_depth=10, src_transfer=TRANSFER_LINEAR,
              dst_transfer=TRANSFER_PQ, src_matrix=MATRIX_GBR, dst_matrix=MATRIX_BT2020NC, src_primaries=9,
              dst_primaries=9, full_range=0, chroma=CHROMA_420, resampler=1, stats=None) -> H2YDesc:
    d = H2YDesc()
    d.width, d.height = width, height
    d.in_sample_type = sample
    d.src_bit_depth, d.dst_bit_depth = src_depth, dst_depth
    d.src_transfer, d.dst_transfer = src_transfer, dst_transfer
    d.src_matrix, d.dst_matrix = src_matrix, dst_matrix
    d.src_primaries, d.dst_primaries = src_primaries, dst_primaries
    d.dst_full_range = full_range
    d.dst_chroma_format_idc = chroma
    d.chroma_resampler_type = resampler
    if stats is not None:
        d.stats_override = 1
        for c in range(3):
            d.floor[c], d.ceiling[c] = int(stats[c][0]), int(stats[c][1])
    return d


def frame_samples(d: H2YDesc) -> int:
    n = d.width * d.height
    nc = (d.width >> 1) * (d.height >> 1) if d.dst_chroma_format_idc == CHROMA_420 else n
    return n + 2 * nc


def _np_dtype(sample):
    return np.float32 if sample == SAMPLE_F32 else np.uint16


def _plane_ptrs(planes, sample):
    arr = (C.c_void_p * 3)()
    keep = []
    for c in range(3):
        p = np.ascontiguousarray(planes[c], dtype=_np_dtype(sample))
        keep.append(p)
        arr[c] = p.ctypes.data
    return arr, keep


def build_oracle(ref: bool = False) -> None:
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", HERE, "--no-print-directory"] + targets, check=True,
                   stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self, build: bool = True):
        path = os.path.join(HERE, "liboracle.so")
        if build and (not os.path.exists(path) or
                      os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "h2y_oracle.c"))):
            build_oracle()
        self.lib = L = C.CDLL(path)
        L.h2y_oracle_pq10000_r.restype = C.c_float
        L.h2y_oracle_pq10000_r.argtypes = [C.c_float]
        L.h2y_oracle_convert_frame.restype = C.c_int
        L.h2y_oracle_convert_frame.argtypes = [C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.c_void_p]
        L.h2y_oracle_frame_bytes.restype = C.c_size_t
        L.h2y_oracle_frame_bytes.argtypes = [C.POINTER(H2YDesc)]
        L.h2y_oracle_sub420_box.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.h2y_oracle_sub420_fir.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        L.h2y_oracle_synth_plane_f32.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.h2y_oracle_f32_to_f16.restype = C.c_uint16
        L.h2y_oracle_f32_to_f16.argtypes = [C.c_float]
        L.h2y_oracle_stats_f32.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.h2y_oracle_matrix_convert.restype = C.c_int
        L.h2y_oracle_matrix_convert.argtypes = [C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                                C.c_int, C.POINTER(C.c_void_p)]

        L.h2y_oracle_matrix_inverse.restype = C.c_int
        L.h2y_oracle_matrix_inverse.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        self._inverse_fn = L.h2y_oracle_matrix_inverse
        L.h2y_oracle_up444.restype = None
        L.h2y_oracle_up444.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint]

    def pq(self, x: float) -> float:
        return float(self.lib.h2y_oracle_pq10000_r(float(x)))

    def up444(self, src: np.ndarray, width: int, height: int, algorithm: int, min_cv: int, max_cv: int, fill: int = 0) -> np.ndarray:
        """Subsample420to444 on a (height//2, width//2) plane -> (height, width); samples the function never writes
        (odd sizes) keep `fill`."""
        src = np.ascontiguousarray(src, dtype=np.uint16).reshape(height >> 1, width >> 1)
        dst = np.full((height, width), fill, dtype=np.uint16)
        self.lib.h2y_oracle_up444(dst.ctypes.data, src.ctypes.data, width, height, algorithm, min_cv, max_cv)
        return dst

    def matrix_inverse(self, width, height, in_depth, in_full_range, in_matrix, out_depth, planes):
        """matrix_inverse() on three U16 4:4:4 planes (Y, Cb/Dz, Cr/Dx) -> (G, B, R) planes."""
        n = width * height
        src = [np.ascontiguousarray(p, dtype=np.uint16).reshape(-1) for p in planes]
        dst = [np.empty(n, dtype=np.uint16) for _ in range(3)]
        ip = (C.c_void_p * 3)(*[p.ctypes.data for p in src])
        op = (C.c_void_p * 3)(*[p.ctypes.data for p in dst])
        rc = self._inverse_fn(width, height, in_depth, in_full_range, in_matrix, out_depth, ip, op)
        if rc != 0:
            raise RuntimeError(f"matrix_inverse rc={rc}")
        return dst

    def convert_frame(self, d: H2YDesc, planes) -> np.ndarray:
        out = np.empty(frame_samples(d), dtype=np.uint16)
        arr, keep = _plane_ptrs(planes, d.in_sample_type)
        rc = self.lib.h2y_oracle_convert_frame(C.byref(d), arr, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"h2y_oracle_convert_frame rc={rc}")
        return out

    def sub420(self, src: np.ndarray, bit_depth: int, fir: bool) -> np.ndarray:
        h, w = src.shape
        src = np.ascontiguousarray(src, dtype=np.uint16)
        dst = np.empty((h >> 1, w >> 1), dtype=np.uint16)
        if fir:
            self.lib.h2y_oracle_sub420_fir(dst.ctypes.data, src.ctypes.data, w, h, 0, (1 << bit_depth) - 1)
        else:
            self.lib.h2y_oracle_sub420_box(dst.ctypes.data, src.ctypes.data, w, h)
        return dst

    def stats_f32(self, planes):
        arr, keep = _plane_ptrs(planes, SAMPLE_F32)
        mm = np.zeros(6, np.float32)
        fl = np.zeros(3, np.int32)
        ce = np.zeros(3, np.int32)
        self.lib.h2y_oracle_stats_f32(arr, keep[0].size, mm.ctypes.data, fl.ctypes.data, ce.ctypes.data)
        return mm, fl, ce

    def matrix_convert(self, d: H2YDesc, planes, floor, ceil, tmp_depth) -> np.ndarray:
        n = d.width * d.height
        arr, keep = _plane_ptrs(planes, SAMPLE_F32 if d.in_sample_type != SAMPLE_U16 else SAMPLE_U16)
        out = np.empty((3, n), np.uint16)
        outp = (C.c_void_p * 3)(*[out[c].ctypes.data for c in range(3)])
        fl = np.asarray(floor, np.int32)
        ce = np.asarray(ceil, np.int32)
        rc = self.lib.h2y_oracle_matrix_convert(C.byref(d), arr, fl.ctypes.data, ce.ctypes.data, tmp_depth, outp)
        if rc != 0:
            raise RuntimeError(f"h2y_oracle_matrix_convert rc={rc}")
        return out

    def synth_frame(self, width: int, height: int, frame: int = 0, f16: bool = False):
        """SURVEY 8c/8d synthetic frame: one LCG stream over planes G,B,R, seed
        12345+frame, values k/2^24, then 0.0 and 1.0 planted at [0],[1]."""
        n = width * height
        st = C.c_uint32(12345 + frame)
        planes = []
        for _ in range(3):
            p = np.empty(n, np.float32)
            self.lib.h2y_oracle_synth_plane_f32(p.ctypes.data, n, C.byref(st))
            planes.append(p)
        if f16:
            planes = [p.astype(np.float16).astype(np.float32) for p in planes]  # RNE, same as h2y_oracle_f32_to_f16
        for p in planes:
            p[0], p[1] = 0.0, 1.0
        if f16:
            return [p.astype(np.float16).view(np.uint16) for p in planes]
        return planes


class Ref:
    """The reference's own object code. Exists only where oracle/_ref was built."""

    def __init__(self, build: bool = True):
        path = os.path.join(HERE, "_ref", "libh2y_ref.so")
        if build and not os.path.exists(path) and os.path.isdir(REF_SRC):
            build_oracle(ref=True)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path)
        L.h2y_ref_pq10000_r.restype = C.c_float
        L.h2y_ref_pq10000_r.argtypes = [C.c_float]
        L.h2y_ref_convert_frame.restype = C.c_int
        L.h2y_ref_convert_frame.argtypes = [C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
        L.h2y_ref_sub420.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.h2y_ref_matrix_inverse.restype = C.c_int
        L.h2y_ref_matrix_inverse.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        self._inverse_fn = L.h2y_ref_matrix_inverse
        L.h2y_ref_up444.restype = C.c_int
        L.h2y_ref_up444.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint]

    def pq(self, x: float) -> float:
        return float(self.lib.h2y_ref_pq10000_r(float(x)))

    def up444(self, src: np.ndarray, width: int, height: int, algorithm: int, min_cv: int, max_cv: int, fill: int = 0) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint16).reshape(height >> 1, width >> 1)
        dst = np.full((height, width), fill, dtype=np.uint16)
        self.lib.h2y_ref_up444(src.ctypes.data, dst.ctypes.data, width, height, algorithm, min_cv, max_cv)
        return dst

    def convert_frame(self, d: H2YDesc, planes, want_444: bool = False):
        out = np.empty(frame_samples(d), dtype=np.uint16)
        arr, keep = _plane_ptrs(planes, d.in_sample_type)
        t444 = np.empty((3, d.width * d.height), np.uint16) if want_444 else None
        rc = self.lib.h2y_ref_convert_frame(C.byref(d), arr, out.ctypes.data,
                                            t444.ctypes.data if want_444 else None)
        if rc != 0:
            raise RuntimeError(f"h2y_ref_convert_frame rc={rc}")
        return (out, t444) if want_444 else out

    def sub420(self, src: np.ndarray, bit_depth: int, fir: bool) -> np.ndarray:
        h, w = src.shape
        src = np.ascontiguousarray(src, dtype=np.uint16)
        dst = np.empty((h >> 1, w >> 1), dtype=np.uint16)
        self.lib.h2y_ref_sub420(src.ctypes.data, dst.ctypes.data, w, h, bit_depth, 1 if fir else 0)
        return dst

    def matrix_inverse(self, width, height, in_depth, in_full_range, in_matrix, out_depth, planes):
        """matrix_inverse() on three U16 4:4:4 planes (Y, Cb/Dz, Cr/Dx) -> (G, B, R) planes."""
        n = width * height
        src = [np.ascontiguousarray(p, dtype=np.uint16).reshape(-1) for p in planes]
        dst = [np.empty(n, dtype=np.uint16) for _ in range(3)]
        ip = (C.c_void_p * 3)(*[p.ctypes.data for p in src])
        op = (C.c_void_p * 3)(*[p.ctypes.data for p in dst])
        rc = self._inverse_fn(width, height, in_depth, in_full_range, in_matrix, out_depth, ip, op)
        if rc != 0:
            raise RuntimeError(f"matrix_inverse rc={rc}")
        return dst


def ref_available() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libh2y_ref.so")) or os.path.isdir(REF_SRC)
