"""The C-ABI library loads, exports every symbol include/hdr2yuv_hip.h declares,
and its host-only entry points behave -- no compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import hdr2yuv_amd as h
from hdr2yuv_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "hdr2yuv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h2y_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_whole_header():
    lib = h.load_library()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hdr2yuv_hip.h but not exported"
    assert sorted(api.EXPORTS) == names
    assert lib.h2y_abi_version() == 1


def test_nm_shows_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", h.library_path()], capture_output=True, text=True).stdout
    assert "gfx950" in out and "k_fused" in out


def test_frame_and_plane_bytes():
    d = h.make_desc(3840, 2160, dst_depth=12, dst_matrix=9, resampler=0)
    assert h.frame_bytes(d) == 24883200  # C2
    d = h.make_desc(3840, 2160, dst_depth=16, dst_matrix=11, chroma=h.CHROMA_444)
    assert h.frame_bytes(d) == 49766400  # C3
    d = h.make_desc(7680, 4320, sample=h.SAMPLE_F16, dst_depth=10)
    assert h.frame_bytes(d) == 99532800  # C4
    lib = h.load_library()
    assert lib.h2y_plane_bytes(C.byref(d)) == 7680 * 4320 * 2


@pytest.mark.parametrize("kw,code", [
    (dict(width=64, height=32), 0),
    (dict(width=0, height=32), 1),
    (dict(width=66, height=32, resampler=0), 1),          # box reads 4x4 tiles
    (dict(width=66, height=32, resampler=1), 0),          # FIR: even is enough
    (dict(width=65, height=32, resampler=1), 1),          # 4:2:0 needs even dims
    (dict(width=65, height=33, chroma=3), 0),             # 4:4:4 any size
    (dict(width=64, height=32, dst_depth=7), 1),
    (dict(width=64, height=32, dst_matrix=4), 2),         # reference exit(0)s: can't determine color difference
    (dict(width=64, height=32, src_transfer=1, dst_transfer=16), 0),   # BT.709 (BT.1886) -> PQ: careful tier
    (dict(width=64, height=32, src_transfer=13, dst_transfer=16), 2),  # sRGB: the reference only prints a warning
    (dict(width=64, height=32, src_transfer=8, dst_transfer=9), 2),    # LOG1 as destination: same
    (dict(width=64, height=32, src_transfer=18, dst_transfer=8), 0),   # RHO_GAMMA source: glibc's powf algorithm, restated
    (dict(width=64, height=32, chroma=2), 2),             # 4:2:2 output
    (dict(width=64, height=32, sample=1, src_depth=10, dst_depth=12, src_transfer=16), 1),  # dst depth > src depth
    (dict(width=64, height=32, stats=[(0, 0), (0, 1), (0, 1)]), 1),   # zero range
])
def test_desc_check(kw, code):
    w, hh = kw.pop("width"), kw.pop("height")
    rc, why = h.desc_check(h.make_desc(w, hh, **kw))
    assert rc == code, why


def test_context_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device creation must raise."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(h.H2YError) as e:
        h.Context(0)
    assert e.value.code == 3 and "no CPU path" in str(e.value)


def test_cli_built_and_fails_loudly_without_gpu():
    import torch

    exe = os.path.join(ROOT, "hdr2yuv_amd", "hdr2yuv")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hdr2yuv_amd", "cli"), "--no-print-directory"], check=True)
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--dst_matrix_coeffs" in r.stdout and "--chroma_resampler_type" in r.stdout
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "--synthetic", "0", "--src_pic_width", "64", "--src_pic_height", "32", "--dst_filename",
                            "/tmp/_h2y_never.yuv", "--src_bit_depth", "32", "--dst_bit_depth", "10", "--src_transfer_characteristics", "8", "--dst_transfer_characteristics", "16", "--dst_matrix_coeffs", "9",
                            "--dst_chroma_format_idc", "1", "--dst_video_full_range_flag", "0"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU path" in r.stdout


def test_experiment_variants_need_the_experiment_flag():
    """Timing variants that write wrong bytes (-DH2Y_EXP_NOCOMPUTE, -DH2Y_SKIP_REDO, ...) do not compile into a product
    library by accident: without -DH2Y_EXPERIMENT the preprocessor stops; with it h2y_abi_version() carries
    H2Y_ABI_EXPERIMENT and the Python binding refuses the library."""
    hdr = os.path.join(ROOT, "hdr2yuv_amd", "csrc", "h2y_math.h")
    for flag in ("-DH2Y_EXP_NOCOMPUTE=1", "-DH2Y_HALF_COMPUTE", "-DH2Y_SKIP_REDO", "-DH2Y_EXP_NOSTATS", "-DH2Y_EXP_NOCONFLICT", "-DH2Y_BLOCK_TIMES"):
        r = subprocess.run(["g++", "-E", "-x", "c++", flag, hdr, "-o", os.devnull], capture_output=True, text=True)
        assert r.returncode != 0 and "H2Y_EXPERIMENT" in r.stderr, flag
        r = subprocess.run(["g++", "-E", "-x", "c++", flag, "-DH2Y_EXPERIMENT", hdr, "-o", os.devnull], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
    from hdr2yuv_amd import api

    assert h.load_library().h2y_abi_version() == api.ABI_VERSION  # the in-tree build is a product build
    assert not api.is_experiment_build()
