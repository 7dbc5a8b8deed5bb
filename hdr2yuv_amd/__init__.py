"""hdr2yuv_amd -- MI355X (gfx950) implementation of hdr2yuv's in-memory convert
path (linear RGB/XYZ -> PQ -> colour-difference matrix -> quantise -> 4:2:0).

The product is ``libhdr2yuv_hip.so`` (hand-written HIP kernels behind the C-ABI
in ``include/hdr2yuv_hip.h``).  This package is the thin Python host mirror of
that ABI used by the tests and ``bench.py``: ctypes only, plus torch for device
memory / streams.  There is no CPU fallback: importing works anywhere, creating
a :class:`Context` needs a HIP device and the built library.
"""
from .api import (  # noqa: F401
    CHROMA_420,
    CHROMA_444,
    H2YDesc,
    H2YError,
    MATRIX_BT2020NC,
    MATRIX_BT709,
    MATRIX_GBR,
    MATRIX_Y100,
    MATRIX_Y500,
    MATRIX_YDZDX,
    SAMPLE_F16,
    SAMPLE_F32,
    SAMPLE_U16,
    TRANSFER_LINEAR,
    TRANSFER_PQ,
    Context,
    build_library,
    desc_check,
    frame_bytes,
    library_path,
    load_library,
    make_desc,
    set_library_path,
)
