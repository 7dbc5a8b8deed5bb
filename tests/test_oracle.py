"""CPU tests of the oracle (oracle/h2y_oracle.c): against the reference's own
object code where that exists (container), against the committed golden
vectors everywhere."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_golden_vectors(oracle):
    index = json.load(open(os.path.join(GOLD, "index.json")))
    assert len(index["cases"]) >= 8
    for case in index["cases"]:
        z = np.load(os.path.join(GOLD, case["file"]))
        d = ob.make_desc(**case["desc"])
        got = oracle.convert_frame(d, [z["in0"], z["in1"], z["in2"]])
        assert np.array_equal(got, z["yuv"]), case["file"]
        assert _md5(got) == case["md5"]


def test_known_md5_small_and_1080p(oracle):
    """SURVEY 8c known answers the oracle can reach in about a second (the 4K/8K
    ones are covered on the GPU side and by tests/golden/make_golden.py)."""
    known = json.load(open(os.path.join(GOLD, "known_md5.json")))["cases"]
    for name in ("tiny_64x32_2020_10b_fir", "C1_1080p_709_10b_fir", "C1_1080p_709_10b_box"):
        case = known[name]
        d = ob.make_desc(**case["desc"])
        planes = oracle.synth_frame(d.width, d.height, 0)
        got = oracle.convert_frame(d, planes)
        assert got.nbytes == case["bytes"]
        assert _md5(got) == case["md5"], name


def test_pq_known_points(oracle):
    # PQ10000_r(0) = (float)pow(0.8359375, 78.84375); PQ10000_r(1) = (float)pow(19.6875/19.6875, ..) = 1
    assert np.float32(oracle.pq(0.0)).view(np.uint32) == 0x354436E8
    assert oracle.pq(1.0) == 1.0
    # monotone on a coarse grid
    xs = np.linspace(0, 1, 257, dtype=np.float32)
    vs = np.array([oracle.pq(float(x)) for x in xs])
    assert np.all(np.diff(vs) > 0)


def test_clip_quirks(oracle):
    """video range scale uses 235*D and 240*D, then the .yuv clamp."""
    w, h = 8, 4
    ones = np.ones(w * h, np.float32)
    ones[0] = 0.0
    d = ob.make_desc(w, h, dst_depth=10, dst_matrix=ob.MATRIX_GBR, chroma=ob.CHROMA_444)
    out = oracle.convert_frame(d, [ones, ones, ones])
    # V=1 -> 1*940+64 = 1004 (identity matrix) -> clamped to maxVR 940 for plane 0, 960 for chroma planes
    assert out[1] == 940 and out[w * h + 1] == 960 and out[2 * w * h + 1] == 960
    assert out[0] == 64  # V(0) = 7.3e-7 -> 64.0006 -> 64


RNG_CASES = [(m, dep, ch, res, fr) for m in (ob.MATRIX_BT2020NC, ob.MATRIX_BT709, ob.MATRIX_YDZDX, ob.MATRIX_Y100,
                                              ob.MATRIX_Y500, ob.MATRIX_GBR)
             for dep in (10, 12, 16) for (ch, res) in ((1, 0), (1, 1), (3, 0)) for fr in (0, 1)]


def test_oracle_equals_reference_object_code(oracle, ref):
    """Randomised shapes/depths/matrices/ranges/resamplers: restatement == reference."""
    rng = np.random.default_rng(7)
    for (m, dep, ch, res, fr) in RNG_CASES:
        w = int(rng.integers(2, 24)) * 4
        h = int(rng.integers(1, 12)) * 4
        planes = [rng.uniform(0, 1, w * h).astype(np.float32) for _ in range(3)]
        for p in planes:
            p[0], p[1] = 0.0, 1.0
        d = ob.make_desc(w, h, dst_depth=dep, dst_matrix=m, chroma=ch, resampler=res, full_range=fr)
        assert np.array_equal(oracle.convert_frame(d, planes), ref.convert_frame(d, planes)), (m, dep, ch, res, fr, w, h)


def test_oracle_equals_reference_other_inputs(oracle, ref):
    rng = np.random.default_rng(8)
    w, h = 48, 20
    # half input, nontrivial floor/ceiling, HDR overshoot above 1.0
    planes = [rng.uniform(1.2, 6.9, w * h).astype(np.float16).view(np.uint16) for _ in range(3)]
    d = ob.make_desc(w, h, sample=ob.SAMPLE_F16, dst_depth=12, dst_matrix=ob.MATRIX_BT2020NC, resampler=1)
    assert np.array_equal(oracle.convert_frame(d, planes), ref.convert_frame(d, planes))
    # 16-bit integer input, no transfer change, 16 -> 10 bit shift in write_yuv
    u16 = [rng.integers(0, 65536, w * h).astype(np.uint16) for _ in range(3)]
    for res in (0, 1):
        d = ob.make_desc(w, h, sample=ob.SAMPLE_U16, src_depth=16, dst_depth=10, src_transfer=16, dst_transfer=16,
                         dst_matrix=ob.MATRIX_YDZDX, resampler=res)
        assert np.array_equal(oracle.convert_frame(d, u16), ref.convert_frame(d, u16))
    # integer input WITH a transfer change: the U16 pic_stats ceiling snap feeds the normalisation
    d = ob.make_desc(w, h, sample=ob.SAMPLE_U16, src_depth=16, dst_depth=12, dst_matrix=ob.MATRIX_BT709, resampler=1)
    assert np.array_equal(oracle.convert_frame(d, u16), ref.convert_frame(d, u16))
    # same transfer, float input: raw values go straight to the matrix
    fl = [rng.uniform(0, 900, w * h).astype(np.float32) for _ in range(3)]
    d = ob.make_desc(w, h, dst_depth=10, src_transfer=16, dst_transfer=16, dst_matrix=ob.MATRIX_BT2020NC, resampler=0)
    assert np.array_equal(oracle.convert_frame(d, fl), ref.convert_frame(d, fl))


def test_subsamplers_equal_reference(oracle, ref):
    rng = np.random.default_rng(9)
    for (w, h, depth) in ((16, 8, 10), (52, 36, 12), (128, 20, 16)):
        src = rng.integers(0, 1 << depth, (h, w)).astype(np.uint16)
        for fir in (False, True):
            assert np.array_equal(oracle.sub420(src, depth, fir), ref.sub420(src, depth, fir)), (w, h, depth, fir)


def test_pq_equals_reference(oracle, ref):
    rng = np.random.default_rng(10)
    xs = np.concatenate([rng.uniform(0, 1, 4000), 2.0 ** rng.uniform(-40, 1, 4000), [0.0, 1.0, 2.0, 1e-30]]).astype(np.float32)
    for x in xs:
        assert np.float32(oracle.pq(float(x))).view(np.uint32) == np.float32(ref.pq(float(x))).view(np.uint32)


def test_other_transfer_pairs_equal_reference(oracle, ref):
    """every source/destination transfer pair the reference has code for
    (convert.cpp:1024-1109), including RHO_GAMMA in both directions."""
    rng = np.random.default_rng(12)
    w, h = 48, 16
    pairs = [(16, 8), (8, 18), (18, 8), (1, 16), (8, 1), (16, 18), (18, 16), (6, 15), (14, 8), (8, 14), (1, 6), (18, 1), (15, 16)]
    for (src, dst) in pairs:
        planes = [rng.uniform(0, 1, w * h).astype(np.float32) for _ in range(3)]
        for p in planes:
            p[0], p[1] = 0.0, 1.0
        for (mat, dep, ch, res) in ((ob.MATRIX_BT2020NC, 12, 1, 1), (ob.MATRIX_YDZDX, 16, 3, 0)):
            d = ob.make_desc(w, h, dst_depth=dep, src_transfer=src, dst_transfer=dst, dst_matrix=mat, chroma=ch, resampler=res)
            assert np.array_equal(oracle.convert_frame(d, planes), ref.convert_frame(d, planes)), (src, dst, mat)


def test_matrix_inverse_restatement_equals_reference(oracle, ref):
    """matrix_inverse() restated vs the reference's object code, every branch the compiled
    function can take (matrix 1 = BT.709; 9, 10 (BT.2020) and 11 all take the Y'DzDx equations), bit-depth
    shifts both ways, full and video range, extreme code values."""
    rng = np.random.default_rng(31)
    w, hh = 96, 20
    for mat in (1, 9, 10, 11, 2):
        for ind, outd, full in ((12, 16, 0), (12, 12, 0), (10, 16, 0), (12, 10, 1), (16, 16, 0), (14, 12, 0), (8, 8, 1)):
            planes = [rng.integers(0, 1 << ind, w * hh).astype(np.uint16) for _ in range(3)]
            for p in planes:
                p[:5] = [0, (1 << ind) - 1, 1 << (ind - 1), (1 << (ind - 1)) - 1, 1]
            a = oracle.matrix_inverse(w, hh, ind, full, mat, outd, planes)
            b = ref.matrix_inverse(w, hh, ind, full, mat, outd, planes)
            for c in range(3):
                assert np.array_equal(a[c], b[c]), (mat, ind, outd, full, c)


def test_upsampler_restatement_equals_reference(oracle, ref):
    """Subsample420to444 (convert.cpp:1869-1986) restated vs the compiled function in oracle/_ref
    (only its call site is under #if 0): replication and the FIR pair, every bit depth's clamp, sizes down to the
    smallest where every edge ternary fires, extreme code values."""
    rng = np.random.default_rng(44)
    for (w, h) in ((2, 2), (4, 2), (2, 6), (8, 8), (12, 4), (64, 32), (130, 18), (256, 66)):
        for depth in (8, 10, 12, 16):
            maxcv = (1 << depth) - 1
            src = rng.integers(0, 1 << depth, (h // 2, w // 2)).astype(np.uint16)
            src.flat[: min(4, src.size)] = [0, maxcv, maxcv, 0][: min(4, src.size)]
            for alg in (0, 1):
                for (lo, hi) in ((0, maxcv), (16 << (depth - 8), 240 << (depth - 8))):
                    a = oracle.up444(src, w, h, alg, lo, hi)
                    b = ref.up444(src, w, h, alg, lo, hi)
                    assert np.array_equal(a, b), (w, h, depth, alg, lo, hi, int(np.count_nonzero(a != b)))
