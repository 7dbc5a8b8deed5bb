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
