"""CPU check of the chunk dealing of the loop-form kernels (hdr2yuv_amd/csrc/h2y_walk.h: frame groups,
XCD-aware layout, weighted rounds): tools/walk_check.cpp includes the very header the kernels compile and
verifies, over ~32 000 configurations, that every chunk of every frame goes to exactly one block."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_chunk_is_dealt_exactly_once(tmp_path):
    exe = str(tmp_path / "walk_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "hdr2yuv_amd", "csrc"),
                    os.path.join(ROOT, "tools", "walk_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 bad" in r.stdout


def test_integer_fir_equals_float_fir(tmp_path):
    """The fused FIR kernel runs Subsample444to420_FIR's two stages in integers (fir_h_int / fir_v_int, h2y_math.h)
    for code values up to 14 bits; tools/fir_int_check.cpp compares them with the float forms (the restated
    reference arithmetic) on random, extreme and flat inputs at every depth 8..14."""
    exe = str(tmp_path / "fir_int_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "hdr2yuv_amd", "csrc"),
                    os.path.join(ROOT, "tools", "fir_int_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout
