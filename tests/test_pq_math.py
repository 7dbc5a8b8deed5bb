"""Host build of the device arithmetic (hdr2yuv_amd/csrc/h2y_math.h) against the
reference formula with this machine's libm, through tools/pq_check.cpp.
Sampled here; the exhaustive runs (every non-negative float) are in DESIGN.md."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "pq_check")


@pytest.fixture(scope="module")
def pq_check():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tools", "pq_check.cpp")
    hdr = os.path.join(ROOT, "hdr2yuv_amd", "csrc", "h2y_math.h")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-mfma", "-std=c++17", "-pthread", src, "-o", EXE], check=True)
    return EXE


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True)
    return r.returncode, r.stdout + r.stderr


def test_fast_tier_whole_top_binades(pq_check):
    # every float in [0.25, 2): 3 binades, 25M values
    rc, out = _run(pq_check, "range", "0x3e800000", "0x40000000", "4")
    assert rc == 0, out
    m = re.search(r"mismatches (\d+), slow tier (\d+) .* max fast err (\d+)", out)
    assert int(m.group(1)) == 0
    assert int(m.group(3)) < 1024  # well inside the 4096-ulp ambiguity window
    assert int(m.group(2)) < 25165824 // 20000


def test_fast_tier_sampled_low_binades(pq_check):
    for lo in (0x33800000, 0x36000000, 0x39000000, 0x3C000000):
        rc, out = _run(pq_check, "range", hex(lo), hex(lo + 0x100000), "4")
        assert rc == 0, out


def test_first_tier_every_float(pq_check):
    """The binary32 first tier (pq_t1, h2y_math.h), every float of its domain [2^-24, 1 + 2^-8) and what lies around it (201 M
    values, ~5 s on eight cores): a sample it calls sure must be the reference's float; its t must lie within H2Y_T1_DELTA ulps
    of the reference double's (the tool fails otherwise); outside the domain, and in the few segments whose values pass a power
    of two, every sample is unsure (and its value NaN: the pixel goes to the binary64 tier)."""
    rc, out = _run(pq_check, "t1", "0x33800000", "0x3f808000", "8")
    assert rc == 0, out
    m = re.search(r"wrong (\d+), flagged (\d+) of (\d+)", out)
    assert int(m.group(1)) == 0 and int(m.group(2)) < int(m.group(3)) // 80, out  # ~0.9 % unsure
    for lo, hi in (("0x33000000", "0x33800000"), ("0x3f808000", "0x40800000"), ("0x00000000", "0x00100000"), ("0x7f000000", "0x80100000")):
        rc, out = _run(pq_check, "t1", lo, hi, "8")
        assert rc == 0, out
        m = re.search(r"wrong 0, flagged (\d+) of (\d+)", out)
        assert m and m.group(1) == m.group(2), out


def test_first_tier_pixels_against_the_exact_tiers():
    """tools/t1_check.cpp: random pixels through pq_t1 + pix_matrix_t1 (raw chroma, no clamp: t1_chroma_in_range()) against the
    exact tiers' integers, nine configurations (depths, ranges, both matrices and YDzDx, three input distributions): a pixel
    the first tier settles must carry the exact integers."""
    exe = os.path.join(ROOT, "build", "t1_check")
    src = os.path.join(ROOT, "tools", "t1_check.cpp")
    hdr = os.path.join(ROOT, "hdr2yuv_amd", "csrc", "h2y_math.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-mfma", "-std=c++17", "-pthread", src, "-o", exe], check=True)
    rc, out = _run(exe, "2000000", "8")
    assert rc == 0, out
    assert out.count("wrong 0") == 9, out


def test_out_of_table_goes_to_slow_tier(pq_check):
    for lo, hi in ((0x00000000, 0x00002000), (0x33000000, 0x33002000), (0x40000000, 0x40002000), (0x7F7FF000, 0x7F800001)):
        rc, out = _run(pq_check, "range", hex(lo), hex(hi), "2")
        assert rc == 0, out
        m = re.search(r": (\d+) floats, mismatches 0, slow tier (\d+)", out)
        # +0.0 is the one out-of-table input the fast tier answers itself (a constant: black bars are common)
        assert m and int(m.group(1)) - int(m.group(2)) == (1 if lo == 0 else 0)


def test_full_range_table_of_the_slow_tier(pq_check):
    """pq_build_table_ext / pq_ext_try (h2y_math.h): the binary64 polynomial table over every normal float below 2, which
    pq_slow() reads from global memory before falling back to double-double arithmetic (samples below the LDS tables'
    2^-24; the other samples of a pixel redone as a whole).  Strided over the whole range (the exhaustive run takes 10 s on
    eight cores: `pq_check ext 0x00800000 0x40000000 8 1`), plus every float of two binades, the subnormals' top and both
    ends of the table (subnormals and x >= 2 must take the double-double tier)."""
    for lo, hi, stride in (("0x00800000", "0x40000000", "509"), ("0x20000000", "0x20800000", "1"), ("0x3F000000", "0x3F800000", "1"),
                           ("0x007F0000", "0x00810000", "1"), ("0x3FFF0000", "0x40010000", "1")):
        r = subprocess.run([pq_check, "ext", lo, hi, "8", stride], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "mismatches 0," in r.stdout, r.stdout


def test_slow_tier_sampled(pq_check):
    rc, out = _run(pq_check, "slow", "0x3f000000", "0x3f040000", "4")
    assert rc == 0, out


def test_pow_dd_within_one_ulp_of_libm(pq_check):
    rc, out = _run(pq_check, "pow", "300000")
    assert rc == 0
    m = re.search(r"m1 differ (\d+) / (\d+) .* m2 differ (\d+) .* >1ulp: (\d+)", out)
    assert int(m.group(4)) == 0
    assert int(m.group(1)) < 0.005 * int(m.group(2)) and int(m.group(3)) < 0.005 * int(m.group(2))


def test_other_transfer_functions_vs_libm(pq_check):
    """PQ10000_f, RHO_GAMMA_f/_r, bt1886_f/_r through the double-double pow/log (and glibc's powf algorithm restated for
    RHO_GAMMA_f's inner power): equal to libm on every sampled input."""
    rc, out = _run(pq_check, "tf", "400000")
    assert rc == 0, out
    m = re.search(r"PQ_f (\d+), RHO_f (\d+), RHO_r (\d+), BT1886_f (\d+), BT1886_r (\d+)", out)
    assert [int(m.group(i)) for i in (1, 2, 3, 4, 5)] == [0, 0, 0, 0, 0]


def test_powf_restatement_equals_libm_on_all_of_zero_to_one(pq_check):
    """powf25() = glibc's powf algorithm (x86-64 FMA build) for x = 25.0f, against this machine's powf over EVERY float of
    [0, 1] (1 065 353 217 values, a few seconds on eight threads) and strided samples of everything else: above 1 up to
    +inf (overflow), every negative float down to -inf (underflow, subnormal results), NaNs."""
    for args in (("0", "0x3f800001", "8"), ("0x3f800000", "0x7f800001", "8", "37"), ("0x80000000", "0xff800001", "8", "41"),
                 ("0x7f800000", "0x7fc00010", "2"), ("0xff800000", "0xffc00010", "2")):
        rc, out = _run(pq_check, "powf", *args)
        assert rc == 0 and " mismatches 0" in out, out


@pytest.mark.parametrize("fn,lo,hi,stride", [(2, "0x39800000", "0x3f800010", 1),   # PQ10000_f on [2^-12, 1]
                                             (3, "0x33800000", "0x40000000", 3),   # bt1886_f (x^2.4) on [2^-24, 2)
                                             (4, "0x33800000", "0x40000000", 3),   # bt1886_r
                                             (5, "0x33800000", "0x40000000", 3),   # RHO_GAMMA_r
                                             (6, "0x3f800000", "0x42000000", 1)])  # RHO_GAMMA_f's outer stage over every P in [1, 32)
def test_other_transfer_tables_equal_libm(pq_check, fn, lo, hi, stride):
    """SURVEY 8f.2 fast tier: the table tier of every other transfer function (tfn_build_table / tfn_fast) against this
    machine's libm over its table's domain -- whatever the tier answers (does not flag for the careful tier) must be the
    reference's float; and it must answer nearly everything (the exhaustive stride-1 runs of functions 3-5 take 3 s more:
    DESIGN.md)."""
    rc, out = _run(pq_check, "tfx", str(fn), lo, hi, "8", str(stride))
    assert rc == 0 and " mismatches 0," in out, out
    m = re.search(r"slow tier (\d+) \(([\d.]+)%\)", out)
    assert float(m.group(2)) < 0.01, out
