"""ctypes mirror of include/hdr2yuv_hip.h (the drop-in C-ABI).

Names follow the reference: a *picture* has three planar planes in the order
0=G/Y, 1=B/Cb(Z/Dz), 2=R/Cr(X/Dx) (hdr.h:359-392); the descriptor carries the
``--src_*`` / ``--dst_*`` attributes of the reference's command line
(hdr2yuv.cpp:73-263).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SAMPLE_U16, SAMPLE_F32, SAMPLE_F16 = 1, 2, 3
CHROMA_420, CHROMA_444 = 1, 3
TRANSFER_LINEAR, TRANSFER_PQ = 8, 16
MATRIX_GBR, MATRIX_BT709, MATRIX_BT2020NC, MATRIX_YDZDX, MATRIX_Y500, MATRIX_Y100 = 0, 1, 9, 11, 12, 13

H2Y_OK, H2Y_EINVAL, H2Y_EUNSUPPORTED, H2Y_EHIP, H2Y_ENOMEM = 0, 1, 2, 3, 4

EXPORTS = [
    "h2y_abi_version", "h2y_frame_bytes", "h2y_plane_bytes", "h2y_desc_check", "h2y_ctx_create", "h2y_ctx_destroy",
    "h2y_last_error", "h2y_ctx_set_stream", "h2y_convert_frame", "h2y_convert_batch", "h2y_convert_batch_enqueue",
    "h2y_batch_finish", "h2y_pic_stats", "h2y_matrix_convert", "h2y_subsample_420", "h2y_last_kernel_ms", "h2y_last_kernel_name", "h2y_last_kernel_variant",
    "h2y_matrix_inverse", "h2y_upsample_444", "h2y_inverse_420", "h2y_inverse_frame", "h2y_ctx_set_option", "h2y_stream_open", "h2y_stream_input", "h2y_stream_submit", "h2y_stream_output", "h2y_stream_close",
]


class H2YError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hdr2yuv_hip error {code}: {msg}")
        self.code = code


class H2YDesc(C.Structure):
    """h2y_desc, include/hdr2yuv_hip.h."""

    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("in_sample_type", C.c_int32),
        ("src_bit_depth", C.c_int32), ("dst_bit_depth", C.c_int32),
        ("src_transfer", C.c_int32), ("dst_transfer", C.c_int32),
        ("src_matrix", C.c_int32), ("dst_matrix", C.c_int32),
        ("src_primaries", C.c_int32), ("dst_primaries", C.c_int32),
        ("dst_full_range", C.c_int32), ("dst_chroma_format_idc", C.c_int32),
        ("chroma_resampler_type", C.c_int32), ("stats_override", C.c_int32),
        ("floor", C.c_int32 * 3), ("ceiling", C.c_int32 * 3),
    ]


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


_LIB_PATH = os.path.join(_HERE, "libhdr2yuv_hip.so")


def library_path() -> str:
    return _LIB_PATH


ABI_VERSION = 1
ABI_EXPERIMENT = 0x40000000  # H2Y_ABI_EXPERIMENT: a -DH2Y_EXPERIMENT build (timing variants, may write wrong bytes)
_ALLOW_EXPERIMENT = False


def set_library_path(path: str, allow_experiment: bool = False) -> None:
    """Load another build of the same library (tuning experiments: bench.py --lib, tools/).  Explicit, never from the
    environment; must be called before the first load_library().  A timing-experiment build (h2y_abi_version() carries
    H2Y_ABI_EXPERIMENT) is refused unless allow_experiment."""
    global _LIB_PATH, _ALLOW_EXPERIMENT
    if _LIB is not None:
        raise RuntimeError("the library is already loaded")
    _LIB_PATH = os.path.abspath(path)
    _ALLOW_EXPERIMENT = allow_experiment


def is_experiment_build() -> bool:
    return bool(load_library().h2y_abi_version() & ABI_EXPERIMENT)


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the kernels + shim, in-tree."""
    args = ["make", "-j4", "-C", os.path.join(_HERE, "csrc"), "--no-print-directory"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    return library_path()


def load_library():
    """Load libhdr2yuv_hip.so. Raises if it has not been built: there is no
    Python or CPU implementation to fall back to."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7; load it
        # first so that our library binds to the same copy (two runtimes in one
        # process cannot both own the device).  Plumbing only -- not required by the
        # library itself (the C++ CLI links the system runtime).
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run hdr2yuv_amd.build_library() (needs hipcc)")
    L = C.CDLL(path)
    L.h2y_abi_version.restype = C.c_int
    ver = L.h2y_abi_version()
    if ver & ABI_EXPERIMENT and not _ALLOW_EXPERIMENT:
        raise RuntimeError(f"{path} is a timing-experiment build (-DH2Y_EXPERIMENT): its kernels may write wrong bytes; refused")
    if ver & ~ABI_EXPERIMENT != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {ver & ~ABI_EXPERIMENT}, this binding is for {ABI_VERSION}")
    L.h2y_frame_bytes.restype = C.c_size_t
    L.h2y_frame_bytes.argtypes = [C.POINTER(H2YDesc)]
    L.h2y_plane_bytes.restype = C.c_size_t
    L.h2y_plane_bytes.argtypes = [C.POINTER(H2YDesc)]
    L.h2y_desc_check.restype = C.c_int
    L.h2y_desc_check.argtypes = [C.POINTER(H2YDesc), C.POINTER(C.c_char_p)]
    L.h2y_ctx_create.restype = C.c_int
    L.h2y_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.h2y_ctx_destroy.restype = None
    L.h2y_ctx_destroy.argtypes = [C.c_void_p]
    L.h2y_last_error.restype = C.c_char_p
    L.h2y_last_error.argtypes = [C.c_void_p]
    L.h2y_ctx_set_option.restype = C.c_int
    L.h2y_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.h2y_ctx_set_stream.restype = C.c_int
    L.h2y_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.h2y_convert_frame.restype = C.c_int
    L.h2y_convert_frame.argtypes = [C.c_void_p, C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.c_void_p]
    for name in ("h2y_convert_batch", "h2y_convert_batch_enqueue"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(H2YDesc), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.h2y_batch_finish.restype = C.c_int
    L.h2y_batch_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.h2y_pic_stats.restype = C.c_int
    L.h2y_pic_stats.argtypes = [C.c_void_p, C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_float),
                                C.POINTER(C.c_int32)]
    L.h2y_matrix_convert.restype = C.c_int
    L.h2y_matrix_convert.argtypes = [C.c_void_p, C.POINTER(H2YDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.h2y_subsample_420.restype = C.c_int
    L.h2y_subsample_420.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.h2y_matrix_inverse.restype = C.c_int
    L.h2y_matrix_inverse.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.h2y_upsample_444.restype = C.c_int
    L.h2y_upsample_444.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    L.h2y_inverse_420.restype = C.c_int
    L.h2y_inverse_420.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.h2y_inverse_frame.restype = C.c_int
    L.h2y_inverse_frame.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.h2y_stream_open.restype = C.c_int
    L.h2y_stream_open.argtypes = [C.c_void_p, C.POINTER(H2YDesc), C.c_int]
    L.h2y_stream_input.restype = C.c_int
    L.h2y_stream_input.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.h2y_stream_submit.restype = C.c_int
    L.h2y_stream_submit.argtypes = [C.c_void_p]
    L.h2y_stream_output.restype = C.c_int
    L.h2y_stream_output.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint16))]
    L.h2y_stream_close.restype = C.c_int
    L.h2y_stream_close.argtypes = [C.c_void_p]
    L.h2y_last_kernel_name.restype = C.c_char_p
    L.h2y_last_kernel_name.argtypes = [C.c_void_p]
    L.h2y_last_kernel_variant.restype = C.c_char_p
    L.h2y_last_kernel_variant.argtypes = [C.c_void_p]
    L.h2y_last_kernel_ms.restype = C.c_int
    L.h2y_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    _LIB = L
    return L


def frame_bytes(d: H2YDesc) -> int:
    return int(load_library().h2y_frame_bytes(C.byref(d)))


def desc_check(d: H2YDesc):
    why = C.c_char_p()
    rc = load_library().h2y_desc_check(C.byref(d), C.byref(why))
    return rc, (why.value or b"").decode()


def _np_dtype(sample):
    return np.float32 if sample == SAMPLE_F32 else np.uint16


class Context:
    """h2y_ctx: one per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.h2y_ctx_create(device, C.byref(h))
        if rc != H2Y_OK:
            raise H2YError(rc, (self.lib.h2y_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.h2y_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != H2Y_OK:
            raise H2YError(rc, (self.lib.h2y_last_error(self.h) or b"").decode())

    def set_option(self, name: str, value) -> None:
        """h2y_ctx_set_option: "t1", "groups", "cols8", "balance", "fir" (tuning / test knobs; output bytes never change)."""
        self._check(self.lib.h2y_ctx_set_option(self.h, name.encode(), str(value).encode()))

    def set_stream(self, hip_stream_ptr: int | None):
        self._check(self.lib.h2y_ctx_set_stream(self.h, C.c_void_p(hip_stream_ptr or 0)))

    # -- host buffers: the reference's whole pic_stats..write_yuv sequence ----
    def convert_frame(self, d: H2YDesc, planes: Sequence[np.ndarray]) -> np.ndarray:
        dt = _np_dtype(d.in_sample_type)
        keep = [np.ascontiguousarray(p, dtype=dt) for p in planes]
        n = d.width * d.height
        for p in keep:
            if p.size != n:
                raise ValueError("plane size does not match width*height")
        arr = (C.c_void_p * 3)(*[p.ctypes.data for p in keep])
        out = np.empty(frame_bytes(d) // 2, dtype=np.uint16)
        self._check(self.lib.h2y_convert_frame(self.h, C.byref(d), arr, out.ctypes.data))
        return out

    # -- device buffers (torch tensors or raw pointers) -----------------------
    @staticmethod
    def _ptr(x) -> int:
        return int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x)

    def _batch_arrays(self, frames_in, frames_out):
        n = len(frames_out)
        ins = (C.c_void_p * (3 * n))()
        outs = (C.c_void_p * n)()
        for f in range(n):
            for c in range(3):
                ins[3 * f + c] = self._ptr(frames_in[f][c])
            outs[f] = self._ptr(frames_out[f])
        return n, ins, outs

    def convert_batch(self, d: H2YDesc, frames_in, frames_out) -> None:
        """frames_in[f] = three device planes (G,B,R); frames_out[f] = device .yuv frame."""
        n, ins, outs = self._batch_arrays(frames_in, frames_out)
        self._check(self.lib.h2y_convert_batch(self.h, C.byref(d), n, ins, outs))

    def convert_batch_enqueue(self, d: H2YDesc, frames_in, frames_out) -> None:
        n, ins, outs = self._batch_arrays(frames_in, frames_out)
        self._keep = (getattr(self, "_keep", ()) + ((ins, outs),))[-2:]  # up to two batches in flight
        self._check(self.lib.h2y_convert_batch_enqueue(self.h, C.byref(d), n, ins, outs))

    def convert_batch_enqueue_raw(self, d: H2YDesc, n: int, ins, outs) -> None:
        """Pre-built pointer arrays (bench loop: no per-step Python marshalling)."""
        self._check(self.lib.h2y_convert_batch_enqueue(self.h, C.byref(d), n, ins, outs))

    def batch_finish(self) -> int:
        redone = C.c_int(0)
        self._check(self.lib.h2y_batch_finish(self.h, C.byref(redone)))
        return redone.value

    def pic_stats(self, d: H2YDesc, planes):
        arr = (C.c_void_p * 3)(*[self._ptr(p) for p in planes])
        mm = (C.c_float * 6)()
        fc = (C.c_int32 * 6)()
        self._check(self.lib.h2y_pic_stats(self.h, C.byref(d), arr, mm, fc))
        return list(mm), list(fc)

    def matrix_convert(self, d: H2YDesc, planes, out_planes) -> None:
        arr = (C.c_void_p * 3)(*[self._ptr(p) for p in planes])
        outp = (C.c_void_p * 3)(*[self._ptr(p) for p in out_planes])
        self._check(self.lib.h2y_matrix_convert(self.h, C.byref(d), arr, outp))

    def subsample_420(self, width, height, bit_depth, resampler, src, dst) -> None:
        self._check(self.lib.h2y_subsample_420(self.h, width, height, bit_depth, resampler, self._ptr(src), self._ptr(dst)))

    def last_kernel_ms(self):
        ms = C.c_float()
        n = C.c_int()
        self.lib.h2y_last_kernel_ms(self.h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def matrix_inverse(self, width, height, in_depth, in_full_range, in_matrix, out_depth, in_planes, out_planes) -> None:
        """Device U16 4:4:4 planes (Y, Cb/Dz, Cr/Dx) -> device U16 planes (G, B, R)."""
        ip = (C.c_void_p * 3)(*[self._ptr(p) for p in in_planes])
        op = (C.c_void_p * 3)(*[self._ptr(p) for p in out_planes])
        self._check(self.lib.h2y_matrix_inverse(self.h, width, height, in_depth, in_full_range, in_matrix, out_depth, ip, op))

    def upsample_444(self, width, height, algorithm, min_cv, max_cv, src, dst) -> None:
        """Subsample420to444: device U16 (height/2 x width/2) plane -> device U16 (height x width) plane."""
        self._check(self.lib.h2y_upsample_444(self.h, width, height, algorithm, min_cv, max_cv, self._ptr(src), self._ptr(dst)))

    def inverse_420(self, width, height, in_depth, in_full_range, in_matrix, out_depth, algorithm, in_planes, out_planes) -> None:
        """Device U16 4:2:0 planes (Y, Cb/Dz, Cr/Dx) -> device U16 planes (G, B, R): upsample, then matrix_inverse."""
        ip = (C.c_void_p * 3)(*[self._ptr(p) for p in in_planes])
        op = (C.c_void_p * 3)(*[self._ptr(p) for p in out_planes])
        self._check(self.lib.h2y_inverse_420(self.h, width, height, in_depth, in_full_range, in_matrix, out_depth, algorithm, ip, op))

    def inverse_frame(self, width, height, in_chroma, in_depth, in_full_range, in_matrix, out_depth, algorithm, in_planes):
        """Host U16 planes (Y, Cb/Dz, Cr/Dx; 4:4:4 or 4:2:0) -> host U16 planes (G, B, R): the .yuv -> .tiff flow on one frame."""
        import numpy as np

        ins = [np.ascontiguousarray(p, dtype=np.uint16) for p in in_planes]
        outs = [np.empty(width * height, np.uint16) for _ in range(3)]
        ip = (C.c_void_p * 3)(*[p.ctypes.data for p in ins])
        op = (C.c_void_p * 3)(*[p.ctypes.data for p in outs])
        self._check(self.lib.h2y_inverse_frame(self.h, width, height, in_chroma, in_depth, in_full_range, in_matrix, out_depth, algorithm, ip, op))
        return outs

    # ---- host <-> device pipeline -----------------------------------------------------------
    def stream_open(self, d, depth=3) -> None:
        self._check(self.lib.h2y_stream_open(self.h, C.byref(d), depth))
        self._stream_desc = d

    def stream_input(self):
        """The three pinned input planes of the next slot, as numpy views to fill in place."""
        import numpy as np

        ptrs = (C.c_void_p * 3)()
        self._check(self.lib.h2y_stream_input(self.h, ptrs))
        d = self._stream_desc
        n = d.width * d.height
        dt = np.float32 if d.in_sample_type == SAMPLE_F32 else np.uint16
        return [np.ctypeslib.as_array(C.cast(ptrs[c], C.POINTER(C.c_float if dt is np.float32 else C.c_uint16)), shape=(n,)) for c in range(3)]

    def stream_submit(self) -> None:
        self._check(self.lib.h2y_stream_submit(self.h))

    def stream_output(self):
        """The oldest frame in flight (a numpy view of pinned memory, valid until the next stream_output)."""
        import numpy as np

        p = C.POINTER(C.c_uint16)()
        self._check(self.lib.h2y_stream_output(self.h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(frame_bytes(self._stream_desc) // 2,))

    def stream_close(self) -> None:
        self._check(self.lib.h2y_stream_close(self.h))

    def last_kernel_name(self) -> str:
        return (self.lib.h2y_last_kernel_name(self.h) or b"").decode()

    def last_kernel_variant(self) -> str:
        return (self.lib.h2y_last_kernel_variant(self.h) or b"").decode()
