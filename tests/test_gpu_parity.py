"""GPU parity: the HIP path, called through the C-ABI, against the oracle
(oracle/h2y_oracle.c, itself pinned to the reference's object code and to the
SURVEY 8c md5s).  Integer output => the bar is bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import hdr2yuv_amd as h
from oracle import binding as ob

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _to_oracle_desc(d):
    od = ob.H2YDesc()
    for name, _ in h.H2YDesc._fields_:
        v = getattr(d, name)
        if name in ("floor", "ceiling"):
            for c in range(3):
                getattr(od, name)[c] = v[c]
        else:
            setattr(od, name, v)
    return od


def _rand_planes(rng, w, h_, kind, lo=0.0, hi=1.0, plant=True):
    n = w * h_
    planes = []
    for _ in range(3):
        p = rng.uniform(lo, hi, n).astype(np.float32)
        # sprinkle exact zeros, tiny values and values that hit the slow tier's domain edges
        idx = rng.integers(0, n, 8)
        p[idx[:3]] = 0.0
        p[idx[3:5]] = np.float32(2.0 ** -30)
        p[idx[5:]] = np.float32(2.0 ** -24)
        if plant:
            p[0], p[1] = lo, hi
        planes.append(p)
    if kind == h.SAMPLE_F16:
        return [p.astype(np.float16).view(np.uint16) for p in planes]
    return planes


CASES = []
for mat in (h.MATRIX_BT2020NC, h.MATRIX_BT709, h.MATRIX_YDZDX, h.MATRIX_Y100, h.MATRIX_Y500, h.MATRIX_GBR):
    for depth in (10, 12, 16):
        for (chroma, res) in ((h.CHROMA_420, 0), (h.CHROMA_420, 1), (h.CHROMA_444, 0)):
            CASES.append((mat, depth, chroma, res))


@pytest.mark.parametrize("mat,depth,chroma,res", CASES)
def test_frame_matches_oracle_f32(ctx, oracle, mat, depth, chroma, res):
    rng = np.random.default_rng(1000 + mat * 31 + depth * 7 + chroma + res)
    w, hh = 136, 52
    for full in (0, 1):
        d = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res, full_range=full)
        planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
        got = ctx.convert_frame(d, planes)
        want = oracle.convert_frame(_to_oracle_desc(d), planes)
        assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} samples differ"


@pytest.mark.parametrize("w,hh", [(4, 4), (8, 4), (64, 32), (132, 12), (260, 36), (1028, 8), (2052, 4)])
@pytest.mark.parametrize("res", [0, 1])
def test_ragged_sizes_420(ctx, oracle, w, hh, res):
    rng = np.random.default_rng(w * 131 + hh)
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=res)
    planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
    got = ctx.convert_frame(d, planes)
    want = oracle.convert_frame(_to_oracle_desc(d), planes)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,hh", [(1, 1), (3, 5), (7, 2), (130, 3), (33, 33), (66, 10)])
def test_odd_sizes_444_and_fir(ctx, oracle, w, hh):
    rng = np.random.default_rng(w * 17 + hh)
    planes = _rand_planes(rng, w, hh, h.SAMPLE_F32, plant=(w * hh >= 2))
    stats = None if w * hh >= 2 else [(0, 1)] * 3
    d = h.make_desc(w, hh, dst_depth=16, dst_matrix=h.MATRIX_YDZDX, chroma=h.CHROMA_444, stats=stats)
    assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))
    if w % 2 == 0 and hh % 2 == 0:
        d = h.make_desc(w, hh, dst_depth=10, dst_matrix=h.MATRIX_BT709, resampler=1)
        assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))


def test_f16_input(ctx, oracle):
    rng = np.random.default_rng(5)
    w, hh = 256, 64
    for res in (0, 1):
        d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=res)
        planes = _rand_planes(rng, w, hh, h.SAMPLE_F16)
        assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))


def test_normalisation_with_nontrivial_stats(ctx, oracle):
    """floor/ceiling other than 0/1: (x - offset) / range in binary32 with an IEEE divide."""
    rng = np.random.default_rng(11)
    w, hh = 128, 32
    planes = [rng.uniform(2.3, 9.7, w * hh).astype(np.float32), rng.uniform(-3.9, 5.2, w * hh).astype(np.float32),
              rng.uniform(0.0, 3.999, w * hh).astype(np.float32)]
    # keep normalised values non-negative: floor is (int)min which truncates toward zero
    planes[1] = np.abs(planes[1]) + np.float32(1.0)
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=1)
    got = ctx.convert_frame(d, planes)
    want = oracle.convert_frame(_to_oracle_desc(d), planes)
    assert np.array_equal(got, want)
    # and the same through an explicit override
    d2 = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0, stats=[(2, 9), (1, 6), (0, 3)])
    assert np.array_equal(ctx.convert_frame(d2, planes), oracle.convert_frame(_to_oracle_desc(d2), planes))


def test_same_transfer_passthrough(ctx, oracle):
    """src transfer == dst transfer: no normalisation, no PQ, no scaling (convert.cpp:930,1012)."""
    rng = np.random.default_rng(12)
    w, hh = 64, 16
    planes = [rng.uniform(0, 1000, w * hh).astype(np.float32) for _ in range(3)]
    d = h.make_desc(w, hh, dst_depth=10, src_transfer=h.TRANSFER_PQ, dst_transfer=h.TRANSFER_PQ,
                    dst_matrix=h.MATRIX_BT709, resampler=1)
    assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))


def test_u16_input(ctx, oracle):
    """16-bit integer input, shifted down in write_yuv."""
    rng = np.random.default_rng(13)
    w, hh = 128, 32
    planes = [rng.integers(0, 65536, w * hh).astype(np.uint16) for _ in range(3)]
    for dst_depth, mat in ((10, h.MATRIX_YDZDX), (12, h.MATRIX_BT2020NC), (16, h.MATRIX_BT709)):
        for res in (0, 1):
            d = h.make_desc(w, hh, sample=h.SAMPLE_U16, src_depth=16, dst_depth=dst_depth, src_transfer=h.TRANSFER_PQ,
                            dst_transfer=h.TRANSFER_PQ, dst_matrix=mat, resampler=res)
            assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))
    # with a transfer change the U16 stats heuristics (common.cpp:94-106) feed the normalisation
    d = h.make_desc(w, hh, sample=h.SAMPLE_U16, src_depth=16, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=1)
    assert np.array_equal(ctx.convert_frame(d, planes), oracle.convert_frame(_to_oracle_desc(d), planes))


def test_golden_fixtures(ctx):
    with open(os.path.join(GOLD, "index.json")) as f:
        index = json.load(f)
    assert index["cases"]
    for case in index["cases"]:
        z = np.load(os.path.join(GOLD, case["file"]))
        d = h.make_desc(**case["desc"])
        planes = [z["in0"], z["in1"], z["in2"]]
        got = ctx.convert_frame(d, planes)
        assert np.array_equal(got, z["yuv"]), case["file"]


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


KNOWN = json.load(open(os.path.join(GOLD, "known_md5.json")))


@pytest.mark.parametrize("name", sorted(KNOWN["cases"].keys()))
def test_full_size_known_md5(ctx, oracle, name):
    """BASELINE configs at full size against the md5 of the reference's own .yuv
    (SURVEY 8c), without running the oracle."""
    case = KNOWN["cases"][name]
    d = h.make_desc(**case["desc"])
    planes = oracle.synth_frame(d.width, d.height, 0, f16=(d.in_sample_type == h.SAMPLE_F16))
    got = ctx.convert_frame(d, planes)
    assert got.nbytes == case["bytes"]
    assert _md5(got) == case["md5"]


def test_batch_device_path_and_stats_redo(ctx, oracle):
    """h2y_convert_batch on device buffers: frames 0..3 share floor/ceiling, frame 4
    does not (its maximum is 2.5) so the assumption fails and it is re-run."""
    import torch

    rng = np.random.default_rng(21)
    w, hh = 256, 64
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(6)]
    host[4][1][7] = np.float32(2.5)
    host[5][0][9] = np.float32(1.5)  # (int)1.5 == 1: same stats as the others
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
    nb = h.frame_bytes(d)
    dev_out = [torch.empty(nb // 2, dtype=torch.int16, device="cuda") for _ in host]
    torch.cuda.synchronize()
    fresh = h.Context(0)
    try:
        fresh.convert_batch_enqueue(d, dev_in, dev_out)
        redone = fresh.batch_finish()
        assert redone == 1
        od = _to_oracle_desc(d)
        for f in range(len(host)):
            got = dev_out[f].cpu().numpy().view(np.uint16)
            assert np.array_equal(got, oracle.convert_frame(od, host[f])), f"frame {f}"
        # second batch: the hint is now frame 5's stats; everything but frame 4 passes
        for t in dev_out:
            t.zero_()
        fresh.convert_batch(d, dev_in, dev_out)
        for f in range(len(host)):
            got = dev_out[f].cpu().numpy().view(np.uint16)
            assert np.array_equal(got, oracle.convert_frame(od, host[f])), f"frame {f} (2nd batch)"
        ms, n = fresh.last_kernel_ms()
        assert n >= 1 and ms > 0
    finally:
        fresh.close()


@pytest.mark.parametrize("mode", ["twopass", "fused", "auto"])
def test_batch_fir(oracle, mode):
    """FIR batches through both forms of the FIR path: 11 frames (one fused launch), and 75 frames -- on the
    two-pass form three sub-batches of 32 frames, so the third reuses the first scratch half behind its event."""
    import torch

    rng = np.random.default_rng(22)
    w, hh = 192, 48
    d = h.make_desc(w, hh, dst_depth=10, dst_matrix=h.MATRIX_BT709, resampler=1)
    od = _to_oracle_desc(d)
    fresh = h.Context(0)
    fresh.set_option("fir", mode)
    try:
        for n in (11, 75):
            host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(n)]
            host[n // 2][2][5] = np.float32(2.5)  # one frame breaks the statistics hint of the second round
            dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
            for rnd in range(2):  # second round: the hint is known (first-tier kernels), one frame redone
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                fresh.convert_batch(d, dev_in, dev_out)
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    want = oracle.convert_frame(od, host[f])
                    assert np.array_equal(got, want), f"{mode} n={n} round {rnd} frame {f}: {np.count_nonzero(got != want)} samples differ"
    finally:
        fresh.close()


@pytest.mark.parametrize("kind", ["box12", "fir10", "ydzdx16_444", "f16_box10"])
def test_long_batch_is_split_into_launches(oracle, kind):
    """BASELINE configs[4]'s single-GPU leg in small: one h2y_convert_batch of 330 frames (> 128 per launch: the
    host splits it, per-launch statistics offsets, ticket counters per frame of a group), mixed content (black bars
    on some frames: dense redo lists), one frame in the last launch breaks the statistics hint and is redone."""
    import torch

    rng = np.random.default_rng(330)
    n, w, hh = 330, 256, 64
    if kind == "box12":
        d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    elif kind == "fir10":
        d = h.make_desc(w, hh, dst_depth=10, dst_matrix=h.MATRIX_BT709, resampler=1)
    elif kind == "ydzdx16_444":
        d = h.make_desc(w, hh, dst_depth=16, dst_matrix=h.MATRIX_YDZDX, chroma=h.CHROMA_444)
    else:
        d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    host = []
    for k in range(n):
        planes = [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
        for p in planes:
            if k % 41 == 7:
                p[: 8 * w] = 0.0
            p[8 * w + (13 * k) % 97] = 1.0  # pic_stats ceiling 1 for every frame
        if k == 300:
            planes[1][3 * w + 9] = np.float32(2.75)  # ceiling 2 in this frame only
        if kind == "f16_box10":
            planes = [p.astype(np.float16).view(np.uint16) for p in planes]
        host.append(planes)
    conv = (lambda p: torch.from_numpy(p.view(np.int16)).cuda()) if kind == "f16_box10" else (lambda p: torch.from_numpy(p).cuda())
    dev_in = [[conv(p) for p in fr] for fr in host]
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    fresh = h.Context(0)
    try:
        for rnd in range(3):
            # the bound is 128 frames per frame GROUP of a launch: round 0 with the default groups, round 1 with
            # one group (128 + 128 + 74), round 2 with two (256 + 74 -- more than 128 frames in one launch)
            if rnd:
                fresh.set_option("groups", rnd)
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            fresh.convert_batch_enqueue(d, dev_in, dev_out)
            redone = fresh.batch_finish()
            assert redone == 1  # frame 300 differs from what was assumed (round 0: frame 0's statistics; later: the hint)
            ms, launches = fresh.last_kernel_ms()
            if rnd == 1:
                assert launches >= 3, launches
            assert launches >= 2, launches
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"{kind} round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
    finally:
        fresh.close()


def test_stage_entries(ctx, oracle):
    """pic_stats / matrix_convert / subsample as separate calls, like the reference's main()."""
    import torch

    rng = np.random.default_rng(23)
    w, hh = 128, 32
    planes = _rand_planes(rng, w, hh, h.SAMPLE_F32, lo=0.0, hi=3.7)
    dev = [torch.from_numpy(p).cuda() for p in planes]
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=1)
    mm, fc = ctx.pic_stats(d, dev)
    omm, ofl, oce = oracle.stats_f32(planes)
    assert np.array_equal(np.float32(mm), omm)
    assert fc[0::2] == list(ofl) and fc[1::2] == list(oce)
    for c in range(3):
        d.floor[c], d.ceiling[c] = fc[2 * c], fc[2 * c + 1]
    out = [torch.empty(w * hh, dtype=torch.int16, device="cuda") for _ in range(3)]
    ctx.matrix_convert(d, dev, out)
    want = oracle.matrix_convert(_to_oracle_desc(d), planes, ofl, oce, 12)
    for c in range(3):
        assert np.array_equal(out[c].cpu().numpy().view(np.uint16), want[c])
    for res in (0, 1):
        dst = torch.empty((hh // 2) * (w // 2), dtype=torch.int16, device="cuda")
        ctx.subsample_420(w, hh, 12, res, out[1], dst)
        ref = oracle.sub420(want[1].reshape(hh, w), 12, fir=bool(res))
        assert np.array_equal(dst.cpu().numpy().view(np.uint16).reshape(hh // 2, w // 2), ref)


def test_descriptor_errors(ctx):
    d = h.make_desc(64, 32, dst_matrix=4)  # MATRIX_FCC: the reference exit(0)s (convert.cpp:1196)
    with pytest.raises(h.H2YError) as e:
        ctx.convert_frame(d, [np.zeros(64 * 32, np.float32)] * 3)
    assert e.value.code == 2
    d = h.make_desc(66, 32, resampler=0)  # box needs multiples of 4
    with pytest.raises(h.H2YError) as e:
        ctx.convert_frame(d, [np.zeros(66 * 32, np.float32)] * 3)
    assert e.value.code == 1


def test_cli_writes_reference_bytes(tmp_path):
    """The C++ host program with the reference's flags: 64x32 synthetic frame, 10-bit
    BT.2020nc 4:2:0 FIR -> the md5 SURVEY 8c recorded from the reference binary; a second
    invocation appends (tiff.cpp:440)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hdr2yuv_amd", "hdr2yuv")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "hdr2yuv_amd", "cli"), "--no-print-directory"], check=True)
    out = str(tmp_path / "o.yuv")
    cmd = [exe, "--synthetic", "0", "--src_pic_width", "64", "--src_pic_height", "32", "--dst_filename", out,
           "--src_bit_depth", "32", "--dst_bit_depth", "10", "--src_transfer_characteristics", "8", "--dst_transfer_characteristics", "16",
           "--dst_matrix_coeffs", "9", "--dst_colour_primaries", "9", "--dst_chroma_format_idc", "1",
           "--dst_video_full_range_flag", "0", "--chroma_resampler_type", "1"]
    for n in (1, 2):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        data = open(out, "rb").read()
        assert len(data) == 6144 * n
        assert hashlib.md5(data[-6144:]).hexdigest() == KNOWN["cases"]["tiny_64x32_2020_10b_fir"]["md5"]


def test_f16_table_path_batch(ctx, oracle):
    """Half input through h2y_convert_batch with known floor 0 / ceiling 1 takes the
    16 384-entry table kernel (k_fused_lut16): zero, subnormal halves, 1.0, and
    overshoot values >= 2.0 (which leave the table and go to the careful tier)."""
    import torch

    rng = np.random.default_rng(31)
    w, hh = 256, 64
    frames = []
    for k in range(3):
        planes = [rng.uniform(0, 1, w * hh).astype(np.float16) for _ in range(3)]
        for p in planes:
            p[0], p[1] = 0.0, 1.0
            p[2:8] = np.array([5.96e-8, 6.0e-5, 1.5, 1.999, 1e-3, 0.0], np.float16)
        if k == 1:  # overshoot, still (int)max == 1
            planes[0][100] = np.float16(1.9)
        frames.append([p.view(np.uint16) for p in planes])
    for (mat, depth, chroma, res) in ((h.MATRIX_BT2020NC, 10, h.CHROMA_420, 0), (h.MATRIX_YDZDX, 16, h.CHROMA_444, 0),
                                      (h.MATRIX_BT709, 12, h.CHROMA_420, 1)):
        d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res,
                        stats=[(0, 1)] * 3)
        dev_in = [[torch.from_numpy(p.view(np.int16)).cuda() for p in fr] for fr in frames]
        dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in frames]
        torch.cuda.synchronize()
        ctx.convert_batch(d, dev_in, dev_out)
        od = _to_oracle_desc(d)
        for f in range(len(frames)):
            assert np.array_equal(dev_out[f].cpu().numpy().view(np.uint16), oracle.convert_frame(od, frames[f])), (mat, f)
    # values >= 2.0 with an override: outside the table, careful tier, still exact
    planes = [rng.uniform(0, 4, w * hh).astype(np.float16).view(np.uint16) for _ in range(3)]
    d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0, stats=[(0, 1)] * 3)
    dev_in = [[torch.from_numpy(p.view(np.int16)).cuda() for p in planes]]
    dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda")]
    torch.cuda.synchronize()
    ctx.convert_batch(d, dev_in, dev_out)
    assert np.array_equal(dev_out[0].cpu().numpy().view(np.uint16), oracle.convert_frame(_to_oracle_desc(d), planes))
    # the hint route (no override): first batch measures, second batch assumes 0/1 and uses the table
    fresh = h.Context(0)
    try:
        d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
        dev_in = [[torch.from_numpy(p.view(np.int16)).cuda() for p in fr] for fr in frames]
        for _ in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in frames]
            torch.cuda.synchronize()
            fresh.convert_batch(d, dev_in, dev_out)
            for f in range(len(frames)):
                assert np.array_equal(dev_out[f].cpu().numpy().view(np.uint16), oracle.convert_frame(_to_oracle_desc(d), frames[f]))
    finally:
        fresh.close()


def test_c4_known_md5_through_batch_table_path(ctx, oracle):
    """C4 (8K half input) at full size through the batch entry with the hint route: md5 of SURVEY 8c."""
    import torch

    case = KNOWN["cases"]["C4_8k_f16_2020_10b_box"]
    d = h.make_desc(**case["desc"])
    planes = oracle.synth_frame(d.width, d.height, 0, f16=True)
    dev_in = [[torch.from_numpy(p.view(np.int16)).cuda() for p in planes]]
    fresh = h.Context(0)
    try:
        for _ in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda")]
            torch.cuda.synchronize()
            fresh.convert_batch(d, dev_in, dev_out)
            assert _md5(dev_out[0].cpu().numpy().view(np.uint16)) == case["md5"]
    finally:
        fresh.close()


def test_caller_stream(ctx, oracle):
    """h2y_ctx_set_stream: launches go to the caller's HIP stream (torch's here)."""
    import torch

    rng = np.random.default_rng(41)
    w, hh = 128, 32
    d = h.make_desc(w, hh, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
    st = torch.cuda.Stream()
    fresh = h.Context(0)
    try:
        fresh.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            dev_in = [[torch.from_numpy(p).cuda(non_blocking=False) for p in planes]]
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda")]
            torch.cuda.synchronize()  # the context's stream does not wait for torch's
        st.synchronize()
        fresh.convert_batch(d, dev_in, dev_out)
        assert np.array_equal(dev_out[0].cpu().numpy().view(np.uint16), oracle.convert_frame(_to_oracle_desc(d), planes))
        fresh.set_stream(None)
    finally:
        fresh.close()


@pytest.mark.parametrize("src,dst", [(16, 8), (8, 18), (1, 16), (8, 1), (16, 18), (6, 15), (14, 8), (8, 14), (1, 6), (15, 16), (18, 8), (18, 16),
                                     (18, 1)])
def test_other_transfer_pairs(ctx, oracle, src, dst):
    """PQ10000_f, RHO_GAMMA_f/_r, bt1886_f/_r at the same dispatch point (careful tier).  RHO_GAMMA as the
    source goes through powf(25, V): glibc's algorithm restated (powf25, pinned over every float of [0, 1] on the CPU)."""
    rng = np.random.default_rng(50 + src * 19 + dst)
    w, hh = 96, 24
    planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
    for (mat, depth, chroma, res) in ((h.MATRIX_BT2020NC, 12, h.CHROMA_420, 1), (h.MATRIX_YDZDX, 16, h.CHROMA_444, 0),
                                      (h.MATRIX_BT709, 10, h.CHROMA_420, 0)):
        d = h.make_desc(w, hh, dst_depth=depth, src_transfer=src, dst_transfer=dst, dst_matrix=mat, chroma=chroma, resampler=res)
        got = ctx.convert_frame(d, planes)
        want = oracle.convert_frame(_to_oracle_desc(d), planes)
        assert np.array_equal(got, want), (mat, np.count_nonzero(got != want))


@pytest.mark.parametrize("src,dst", [(16, 8), (8, 1), (1, 16), (16, 1), (18, 16), (8, 18), (16, 18)])
@pytest.mark.parametrize("sample", ["f32", "f16", "u16"])
def test_transfer_pairs_loop_kernel_dark_batches(oracle, src, dst, sample):
    """The generic pairs in the loop-form kernel (k_fused2<...,TFN>: both stages' tables in LDS, what they do not reach
    from the stages' full-range tables in global memory, by scalar loads) on batches -- second round on the statistics
    hint -- of DARK frames: uniform noise cubed, so that 0.4 % of the samples fall below a 2^-24 table floor and 6 % below
    PQ10000_f's 2^-12, plus zeros and a few negatives and values above one.  Half input must not take the LINEAR -> PQ
    table kernel (k_fused_lut16) for another pair, whatever the statistics."""
    import torch

    rng = np.random.default_rng(700 + src * 31 + dst)
    w, hh, n = 256, 64, 4
    host = []
    for k in range(n):
        planes = [(rng.random(w * hh, dtype=np.float32) ** 3).astype(np.float32) for _ in range(3)]
        for p in planes:
            p[0], p[1] = 0.0, 1.0
            idx = rng.integers(2, p.size, 6)
            p[idx[:3]] = 0.0
            if sample == "f32":
                p[idx[3]] = np.float32(-0.01)
                p[idx[4]] = np.float32(1.25)
        if sample == "f16":
            planes = [p.astype(np.float16).view(np.uint16) for p in planes]
        elif sample == "u16":
            planes = [np.minimum(p * 4095.0, 4095.0).astype(np.uint16) for p in planes]
        host.append(planes)
    kind = {"f32": h.SAMPLE_F32, "f16": h.SAMPLE_F16, "u16": h.SAMPLE_U16}[sample]
    for (mat, depth, chroma, res) in ((h.MATRIX_BT2020NC, 12, h.CHROMA_420, 0), (h.MATRIX_BT709, 10, h.CHROMA_420, 1), (h.MATRIX_YDZDX, 12, h.CHROMA_444, 0)):
        d = h.make_desc(w, hh, sample=kind, src_depth=12 if sample == "u16" else 32, dst_depth=depth, src_transfer=src, dst_transfer=dst,
                        dst_matrix=mat, chroma=chroma, resampler=res)
        od = _to_oracle_desc(d)
        want = [oracle.convert_frame(od, fr) for fr in host]
        c = h.Context(0)
        try:
            dev_in = [[torch.from_numpy(np.ascontiguousarray(p).view(np.int16) if p.dtype == np.uint16 else p).cuda() for p in fr] for fr in host]
            for rnd in range(2):
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                c.convert_batch(d, dev_in, dev_out)
                assert c.last_kernel_name() == "k_fused2" and ",TFN" in c.last_kernel_variant(), c.last_kernel_variant()
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    assert np.array_equal(got, want[f]), (sample, src, dst, mat, rnd, f, int(np.count_nonzero(got != want[f])))
        finally:
            c.close()


def test_rho_gamma_source_out_of_range_samples(ctx, oracle):
    """RHO_GAMMA_f on samples outside [0, 1]: negative V gives (25^V - 1) / 24 < 0 and pow(negative, 2.4) = NaN, large V
    overflows powf to infinity: the reference's conversions of those, byte for byte."""
    rng = np.random.default_rng(18)
    w, hh = 64, 16
    planes = [rng.uniform(-0.5, 1.5, w * hh).astype(np.float32) for _ in range(3)]
    planes[0][:6] = [0.0, 1.0, -0.0, 30.0, -40.0, 1e-30]
    for (mat, depth, chroma) in ((h.MATRIX_BT2020NC, 12, h.CHROMA_420), (h.MATRIX_YDZDX, 16, h.CHROMA_444)):
        d = h.make_desc(w, hh, dst_depth=depth, src_transfer=18, dst_transfer=16, dst_matrix=mat, chroma=chroma, resampler=1, stats=[(0, 1)] * 3)
        got = ctx.convert_frame(d, planes)
        want = oracle.convert_frame(_to_oracle_desc(d), planes)
        assert np.array_equal(got, want), (mat, int(np.count_nonzero(got != want)))


def _picture_like(rng, w, hh):
    """Random picture with what real material has and uniform noise has not: black bars (+0.0),
    a saturated patch with unused channels, out-of-gamut negatives, over-range highlights."""
    planes = [rng.uniform(0.0, 1.0, (hh, w)).astype(np.float32) for _ in range(3)]
    bar = max(2, hh // 8)
    for p in planes:
        p[:bar, :] = 0.0
        p[-bar:, :] = 0.0
    planes[0][bar:bar + 6, 8:40] = 0.0           # pure blue/red patch: G unused
    planes[1][bar + 6:bar + 10, 16:24] = -0.0     # negative zero
    planes[2][bar + 10:bar + 12, 4:12] = np.float32(-0.125)
    planes[1][bar + 12:bar + 14, 40:48] = np.float32(1.75)
    planes[0][bar + 14, 3] = np.float32(2.0 ** -26)
    planes[2][bar + 14, 5] = np.float32(1e-41)    # denormal
    return [p.reshape(-1) for p in planes]


@pytest.mark.parametrize("depth,mat,chroma,res", [(12, h.MATRIX_BT2020NC, h.CHROMA_420, 0), (10, h.MATRIX_BT709, h.CHROMA_444, 0),
                                                  (12, h.MATRIX_BT2020NC, h.CHROMA_420, 1), (10, h.MATRIX_YDZDX, h.CHROMA_420, 0),
                                                  (16, h.MATRIX_YDZDX, h.CHROMA_444, 0)])
def test_black_bars_and_out_of_table_samples(ctx, oracle, depth, mat, chroma, res):
    """First-tier kernel: tiles holding +0.0 / negative / >= 2-adjacent samples leave the binary32
    tier through the tile-level exit, the rest through the redo list; bytes still the oracle's."""
    rng = np.random.default_rng(77 + depth)
    w, hh = 264, 80
    d = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res, stats=[(0, 1)] * 3)
    planes = _picture_like(rng, w, hh)
    got = ctx.convert_frame(d, planes)
    want = oracle.convert_frame(_to_oracle_desc(d), planes)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} samples differ"
    # measured statistics (floor -0 -> 0, ceiling 1) and the normalising pipe
    d2 = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res)
    got = ctx.convert_frame(d2, planes)
    want = oracle.convert_frame(_to_oracle_desc(d2), planes)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} samples differ (measured stats)"


@pytest.mark.parametrize("depth,mat,chroma,res", [(12, h.MATRIX_BT2020NC, h.CHROMA_420, 0), (10, h.MATRIX_YDZDX, h.CHROMA_444, 0),
                                                  (12, h.MATRIX_BT709, h.CHROMA_420, 1)])
def test_batch_redo_lists_across_frames(ctx, oracle, depth, mat, chroma, res):
    """One launch over many frames: the first-tier kernel prefetches across frame boundaries and its
    per-wave redo lists hold tiles of several frames (uniform noise: a few per frame; the frames with
    black bars: every tile of the bars)."""
    import torch

    rng = np.random.default_rng(4242 + depth)
    w, hh = 512, 96
    d = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res)
    host = []
    for k in range(10):
        planes = _picture_like(rng, w, hh) if k % 3 == 1 else [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
        for p in planes:
            p[(7 * k) % 64] = 1.0  # pic_stats ceiling 1 for every frame
        host.append(planes)
    dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in fr] for fr in host]
    nb = h.frame_bytes(d)
    dev_out = [torch.zeros(nb // 2, dtype=torch.int16, device="cuda") for _ in host]
    torch.cuda.synchronize()
    ctx.convert_batch(d, dev_in, dev_out)
    od = _to_oracle_desc(d)
    for f in range(len(host)):
        got = dev_out[f].cpu().numpy().view(np.uint16)
        want = oracle.convert_frame(od, host[f])
        assert np.array_equal(got, want), f"frame {f}: {np.count_nonzero(got != want)} samples differ"


@pytest.mark.parametrize("res,depth_out", [(0, 12), (1, 10)])
def test_stream_pipeline(oracle, res, depth_out):
    """frames through the pinned ring (upload / convert / download overlapped) come out in
    order and byte-identical; frames with different statistics follow each other (no speculation here)."""
    rng = np.random.default_rng(99 + res)
    w, hh = 320, 64
    d = h.make_desc(w, hh, dst_depth=depth_out, dst_matrix=h.MATRIX_BT2020NC, resampler=res)
    od = _to_oracle_desc(d)
    frames = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(7)]
    frames[3][1][5] = np.float32(2.5)  # ceiling 2 for this one only
    c = h.Context(0)
    try:
        c.stream_open(d, 3)
        got, inflight = [], 0
        for fr in frames:
            dst = c.stream_input()
            for k in range(3):
                dst[k][:] = fr[k]
            c.stream_submit()
            inflight += 1
            if inflight == 2:
                got.append(c.stream_output().copy())
                inflight -= 1
        while inflight:
            got.append(c.stream_output().copy())
            inflight -= 1
        c.stream_close()
        assert len(got) == len(frames)
        for k, fr in enumerate(frames):
            assert np.array_equal(got[k], oracle.convert_frame(od, fr)), f"frame {k}"
        # the context is usable again afterwards
        assert np.array_equal(c.convert_frame(d, frames[0]), got[0])
    finally:
        c.close()


def test_matrix_inverse(ctx, oracle):
    """h2y_matrix_inverse vs the oracle (itself pinned to the reference's object code)."""
    import torch

    rng = np.random.default_rng(32)
    for (w, hh) in ((256, 64), (67, 9)):
        n = w * hh
        for mat in (1, 9, 11):
            for ind, outd, full in ((12, 16, 0), (12, 12, 0), (10, 16, 0), (12, 10, 1), (16, 16, 0)):
                planes = [rng.integers(0, 1 << ind, n).astype(np.uint16) for _ in range(3)]
                for p in planes:
                    p[:5] = [0, (1 << ind) - 1, 1 << (ind - 1), (1 << (ind - 1)) - 1, 1]
                din = [torch.from_numpy(p.view(np.int16)).cuda() for p in planes]
                dout = [torch.zeros(n, dtype=torch.int16, device="cuda") for _ in range(3)]
                torch.cuda.synchronize()  # the context's stream does not wait for torch's
                ctx.matrix_inverse(w, hh, ind, full, mat, outd, din, dout)
                want = oracle.matrix_inverse(w, hh, ind, full, mat, outd, planes)
                for c in range(3):
                    got = dout[c].cpu().numpy().view(np.uint16)
                    assert np.array_equal(got, want[c]), (w, hh, mat, ind, outd, full, c, int(np.count_nonzero(got != want[c])))
    with pytest.raises(h.H2YError):
        ctx.matrix_inverse(256, 64, 12, 0, 0, 16, din, dout)  # GBR: the reference exits


@pytest.mark.parametrize("name", ["C2_4k_2020_12b_box", "C2_4k_2020_12b_fir", "C3_4k_ydzdx_16b_444"])
def test_full_size_batch_of_frames(ctx, oracle, name):
    """One launch over three full-size frames (prefetch across frame boundaries, redo lists spanning
    frames, FIR sub-batches): frame 0 is the SURVEY 8c known answer, frames 1 and 2 (generator seeds
    12346, 12347) are checked against the oracle."""
    import torch

    case = KNOWN["cases"][name]
    d = h.make_desc(**case["desc"])
    host = [oracle.synth_frame(d.width, d.height, k) for k in range(3)]
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
    nb = h.frame_bytes(d)
    dev_out = [torch.zeros(nb // 2, dtype=torch.int16, device="cuda") for _ in host]
    torch.cuda.synchronize()
    ctx.convert_batch(d, dev_in, dev_out)
    got0 = dev_out[0].cpu().numpy().view(np.uint16)
    assert _md5(got0) == case["md5"]
    od = _to_oracle_desc(d)
    for f in (1, 2):
        got = dev_out[f].cpu().numpy().view(np.uint16)
        want = oracle.convert_frame(od, host[f])
        assert np.array_equal(got, want), f"{name} frame {f}: {np.count_nonzero(got != want)} samples differ"


def test_fuzz_descriptors_against_oracle(ctx, oracle):
    """Seeded sweep over the descriptor space: sizes (all tile / chunk remainders, odd heights, narrow
    widths), sample types, bit depths, matrices, ranges, resamplers, statistics (measured, overridden,
    non-trivial), single frames and small batches -- every kernel variant the dispatcher can pick."""
    import torch

    rng = np.random.default_rng(20261004)
    mats = [h.MATRIX_BT2020NC, h.MATRIX_BT709, h.MATRIX_YDZDX, h.MATRIX_Y100, h.MATRIX_Y500, h.MATRIX_GBR]
    n_cases = 0
    for it in range(300):
        chroma = h.CHROMA_420 if rng.random() < 0.6 else h.CHROMA_444
        res = int(rng.integers(0, 2))
        if chroma == h.CHROMA_420:
            step = 4 if res == 0 else 2
            w = int(rng.integers(1, 90)) * step
            hh = int(rng.integers(1, 40)) * step
        else:
            w = int(rng.integers(1, 300))
            hh = int(rng.integers(1, 70))
        sample = [h.SAMPLE_F32, h.SAMPLE_F32, h.SAMPLE_F16, h.SAMPLE_U16][int(rng.integers(0, 4))]
        depth = int(rng.choice([8, 10, 12, 14, 16]))
        mat = mats[int(rng.integers(0, len(mats)))]
        full = int(rng.integers(0, 2))
        kw = dict(dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res, full_range=full, sample=sample)
        same_transfer = rng.random() < 0.2
        if same_transfer:
            kw.update(src_transfer=h.TRANSFER_PQ, dst_transfer=h.TRANSFER_PQ)
        if sample == h.SAMPLE_U16:
            kw["src_depth"] = int(rng.choice([d for d in (10, 12, 16) if d >= depth] or [16]))
            if kw["src_depth"] < depth:
                kw["dst_depth"] = kw["src_depth"]
        mode = int(rng.integers(0, 4))  # 0: measured stats 0/1, 1: override 0/1, 2: measured with ceiling 2, 3: override (-1, 2)
        if mode == 1:
            kw["stats"] = [(0, 1)] * 3
        if mode == 3:
            kw["stats"] = [(-1, 2)] * 3
        d = h.make_desc(w, hh, **kw)
        n = w * hh
        nframes = 1 if rng.random() < 0.7 else int(rng.integers(2, 5))
        frames = []
        for _ in range(nframes):
            if sample == h.SAMPLE_U16:
                planes = [rng.integers(0, 1 << kw["src_depth"], n).astype(np.uint16) for _ in range(3)]
            else:
                planes = _rand_planes(rng, w, hh, sample, plant=(n >= 2))
                if mode == 2 and n >= 3 and sample == h.SAMPLE_F32:
                    planes[int(rng.integers(0, 3))][2] = np.float32(2.25)
                if n >= 16 and rng.random() < 0.3 and sample == h.SAMPLE_F32:
                    planes[1][5:9] = 0.0
                    planes[2][9] = np.float32(-0.5)
            frames.append(planes)
        od = _to_oracle_desc(d)
        if nframes == 1:
            got = [ctx.convert_frame(d, frames[0])]
        else:
            conv = (lambda p: torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).cuda())
            dev_in = [[conv(np.ascontiguousarray(p)) for p in fr] for fr in frames]
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in frames]
            torch.cuda.synchronize()  # the context's stream does not wait for torch's
            ctx.convert_batch(d, dev_in, dev_out)
            got = [t.cpu().numpy().view(np.uint16) for t in dev_out]
        for f, planes in enumerate(frames):
            want = oracle.convert_frame(od, planes)
            assert np.array_equal(got[f], want), (it, f, w, hh, kw, mode, int(np.count_nonzero(got[f] != want)))
        n_cases += 1
    assert n_cases == 300


def test_tier_steering_on_black_frames(oracle):
    """Exact zeros scattered through a picture (not whole rows of them: those the first tier answers itself, see
    test_rows_of_zeros_stay_on_the_first_tier) make the first-tier kernel redo every tile; the context notices (share
    of redone tiles) and sends the next batches to the binary64-tier kernel, probing the first tier again
    later.  Bytes are the oracle's whichever kernel runs."""
    import torch

    w, hh = 512, 256
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0, stats=[(0, 1)] * 3)
    od = _to_oracle_desc(d)
    rng = np.random.default_rng(5)
    black = [rng.uniform(0, 1, w * hh).astype(np.float32) for _ in range(3)]
    for p in black:
        p[rng.random(p.size) < 0.5] = 0.0  # half of the samples: every tile holds some, no row is all zero
    noise = [rng.uniform(0, 1, w * hh).astype(np.float32) for _ in range(3)]
    want_black, want_noise = oracle.convert_frame(od, black), oracle.convert_frame(od, noise)
    c = h.Context(0)
    try:
        names = []
        for step in range(37):
            planes = black if step < 35 else noise
            dev_in = [[torch.from_numpy(p).cuda() for p in planes] for _ in range(2)]
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()  # the context's stream does not wait for torch's
            c.convert_batch(d, dev_in, dev_out)
            names.append(c.last_kernel_name())
            for t in dev_out:
                assert np.array_equal(t.cpu().numpy().view(np.uint16), want_black if step < 35 else want_noise), step
        assert names[0] == "k_fused_t1"              # first batch: nothing known yet
        assert names[1:33] == ["k_fused2"] * 32      # 32 batches away from the first tier (a probe on dense content is dear)
        assert names[33] == "k_fused_t1"             # probe: still black
        assert names[34:] == ["k_fused2"] * 3        # away again (for 64 batches now), whatever the pictures hold
    finally:
        c.close()


@pytest.mark.parametrize("sample,hi", [(h.SAMPLE_F32, 1.0), (h.SAMPLE_F32, 2.6), (h.SAMPLE_F16, 2.6)])
@pytest.mark.parametrize("res", [0, 1])
def test_rows_of_zeros_with_measured_statistics(oracle, res, sample, hi):
    """The same shortcut on the kernels' other forms: statistics measured by the kernel (round 0: the normalising variant
    with the first frame's floor / ceiling assumed; ceiling 2 keeps it there), half input through the first tier."""
    import torch

    rng = np.random.default_rng(int(17 + res + 10 * hi))
    w, hh, n = 512, 96, 5
    host = []
    for k in range(n):
        planes = [rng.uniform(0.0, hi, w * hh).astype(np.float32) for _ in range(3)]
        for p in planes:
            img = p.reshape(hh, w)
            img[: 12 + 2 * k] = 0.0
            img[hh - 10:] = 0.0
            img[40, 1] = hi
        host.append([p.astype(np.float16).view(np.uint16) for p in planes] if sample == h.SAMPLE_F16 else planes)
    d = h.make_desc(w, hh, sample=sample, dst_depth=10 if res else 12, dst_matrix=h.MATRIX_BT709, resampler=res)
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    conv = (lambda p: torch.from_numpy(p.view(np.int16)).cuda()) if sample == h.SAMPLE_F16 else (lambda p: torch.from_numpy(p).cuda())
    dev_in = [[conv(p) for p in fr] for fr in host]
    c = h.Context(0)
    try:
        if res:
            c.set_option("fir", "fused")
        names = set()
        for rnd in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            names.add(c.last_kernel_variant().split(" ")[0])
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ ({names})"
        assert any(v.startswith("k_fir_fused" if res else "k_fused_t1") for v in names) or (sample == h.SAMPLE_F16 and hi == 1.0), names
    finally:
        c.close()


@pytest.mark.parametrize("res", [0, 1])
def test_rows_of_zeros_stay_on_the_first_tier(oracle, res):
    """Letterbox bars: rows that are +0.0 in every lane of a wave are recognised before the first tier's arithmetic (which
    cannot answer zero) and given the black pixel's code values; they are not flagged, so such pictures stay on k_fused_t1 /
    k_fir_fused.  Bars of full rows, a bar ending inside a wave's row pair, a row that is zero except for one sample and one
    with a -0.0 in it (both must take the ordinary path), all-black frames, measured and assumed statistics."""
    import torch

    rng = np.random.default_rng(91 + res)
    w, hh, n = 512, 512, 6  # (tall enough for the two odd rows' pixels, which the first tier does pass on, to stay below the share -- 0.7 % -- that steers away)
    host = []
    for k in range(n):
        planes = [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
        for c, p in enumerate(planes):
            img = p.reshape(hh, w)
            if k == 5:
                img[:] = 0.0                      # an all-black frame
            else:
                img[: 16 + k] = 0.0               # top bar: 16..20 rows (odd heights end inside a row pair)
                img[hh - 13:] = 0.0               # bottom bar
                img[3, 77] = np.float32(0.25) if c == 1 else img[3, 77]   # one sample in a bar row
                img[5, 300] = np.float32(-0.0) if c == 2 else img[5, 300]  # -0.0 is not the bit pattern the shortcut looks for
            img[40, 1] = 1.0                      # ceiling 1 in every frame but the black one
        host.append(planes)
    d = h.make_desc(w, hh, dst_depth=10 if res else 12, dst_matrix=h.MATRIX_BT2020NC, resampler=res, stats=[(0, 1)] * 3)
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
    c = h.Context(0)
    try:
        if res:
            c.set_option("fir", "fused")  # (frames this small would take the two-pass form by themselves; dense zeros still steer away)
        for rnd in range(3):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            assert c.last_kernel_name() == ("k_fir_fused" if res else "k_fused_t1"), (rnd, c.last_kernel_variant())  # never steered away
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
    finally:
        c.close()


def test_subsampled_minimum_cases(oracle):
    """k_fused_t1 with the assumed floor 0 / ceiling 1 tracks the maximum of every sample but only a
    subsample of the minimum (row 0, columns 0-1 of every 4x2 tile).  Frames where that is not enough must be
    noticed and redone with exact statistics: a sample <= -1 off the subsample (floor -1), a frame whose
    subsample is all >= 1 although other samples are small (floor 0 after all), and both in one batch with
    ordinary frames."""
    import torch

    rng = np.random.default_rng(77)
    w, hh = 256, 64
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    od = _to_oracle_desc(d)
    host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(6)]
    host[2][1][3 * w + 7] = np.float32(-1.5)            # odd row, column 7: never in the subsample
    img = [p.reshape(hh, w) for p in host[4]]
    for p in img:
        p[0::2, 0::4] = np.float32(1.25)                 # the whole subsample >= 1 ...
        p[0::2, 1::4] = np.float32(1.5)
        p[1, 2] = np.float32(0.25)                       # ... but the picture's minimum is not
    dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in fr] for fr in host]
    dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
    torch.cuda.synchronize()  # the context's stream does not wait for torch's
    c = h.Context(0)
    try:
        c.convert_batch(d, dev_in[:2], dev_out[:2])       # establishes the hint floor 0 / ceiling 1
        for t in dev_out:
            t.zero_()
        c.convert_batch_enqueue(d, dev_in, dev_out)
        assert c.last_kernel_name() == "k_fused_t1"
        redone = c.batch_finish()
        assert redone == 2
        for f in range(len(host)):
            got = dev_out[f].cpu().numpy().view(np.uint16)
            assert np.array_equal(got, oracle.convert_frame(od, host[f])), f"frame {f}"
        # and the next batch still runs on a correct hint (the last frame was ordinary)
        c.convert_batch(d, dev_in[:2], dev_out[:2])
        assert np.array_equal(dev_out[1].cpu().numpy().view(np.uint16), oracle.convert_frame(od, host[1]))
    finally:
        c.close()


@pytest.mark.parametrize("kind,n,w,hh", [("f32", 8, 512, 96), ("f32", 16, 1024, 540), ("f32", 12, 256, 36), ("f32", 24, 64, 8),
                                         ("f16", 8, 512, 96), ("u16", 8, 512, 96), ("f32_444", 8, 260 * 4, 66)])
def test_frame_groups(oracle, kind, n, w, hh):
    """Batches whose length is a multiple of 2/4/8: the loop-form kernels split the grid into that many
    groups, each walking every 8th (4th, 2nd) frame (frame_walk).  Frames differ in content, some have
    black bars (dense redo lists), one breaks the statistics assumption (redone afterwards); per-frame
    statistics land in the right frame's slot or the redo would hit the wrong frame."""
    import torch

    rng = np.random.default_rng(9000 + n + w)
    if kind == "f16":
        d = h.make_desc(w, hh, sample=h.SAMPLE_F16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    elif kind == "u16":
        d = h.make_desc(w, hh, sample=h.SAMPLE_U16, src_depth=16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0,
                        src_transfer=16, dst_transfer=16)
    elif kind == "f32_444":
        d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_YDZDX, chroma=h.CHROMA_444)
    else:
        d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    host = []
    for k in range(n):
        if kind == "u16":
            planes = [rng.integers(0, 65536, w * hh, dtype=np.uint16) for _ in range(3)]
        else:
            planes = [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
            for p in planes:
                p[(11 * k) % 50] = 1.0
                if k % 5 == 2:  # bars
                    p[: (hh // 8) * w] = 0.0
            if k == n - 3:
                planes[2][w + 1] = np.float32(2.25)  # ceiling 2 in this frame only
            if kind == "f16":
                planes = [p.astype(np.float16).view(np.uint16) for p in planes]
        host.append(planes)
    as_dev = (lambda p: torch.from_numpy(np.ascontiguousarray(p).view(np.int16)).cuda()) if kind in ("f16", "u16") else \
             (lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda())
    dev_in = [[as_dev(p) for p in fr] for fr in host]
    fresh = h.Context(0)
    try:
        od = _to_oracle_desc(d)
        want = [oracle.convert_frame(od, fr) for fr in host]
        for rnd in range(2):  # second round: hint known, the first-tier / table kernels run
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            fresh.convert_batch(d, dev_in, dev_out)
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
    finally:
        fresh.close()


@pytest.mark.parametrize("balance", ["0x55,1.07", "0xAA,1.4", "0x01,1.5", "0xFE,1.2", "off"])
def test_weighted_rounds(oracle, balance):
    """The loop-form kernels deal a frame's chunks in two parts -- all blocks, then the blocks on the fast
    XCDs only (frame_walk) -- with weights the host derives from block finish times.  Here the weights are
    fixed through h2y_ctx_set_option("balance"), extreme ones included; frames large enough for the full grid
    (so that the XCD layout applies) and batch lengths with 1, 2 and 8 frame groups; every chunk must be done
    exactly once."""
    import torch

    rng = np.random.default_rng(515)
    w, hh = 2048, 1024  # 262 144 tiles: 256 chunks of 1024 (k_fused_t1), 512 of 512 (k_fused2)
    fresh = h.Context(0)
    fresh.set_option("balance", balance)
    try:
        for (n, depth, mat, chroma) in ((3, 12, h.MATRIX_BT2020NC, h.CHROMA_420), (8, 12, h.MATRIX_BT2020NC, h.CHROMA_420),
                                        (2, 16, h.MATRIX_YDZDX, h.CHROMA_444)):
            d = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=0)
            host = []
            for k in range(n):
                planes = [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
                for p in planes:
                    p[k] = 1.0
                host.append(planes)
            dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
            od = _to_oracle_desc(d)
            want = [oracle.convert_frame(od, fr) for fr in host]
            for rnd in range(2):
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                fresh.convert_batch(d, dev_in, dev_out)
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    assert np.array_equal(got, want[f]), f"{balance} n={n} round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
                if rnd == 1:
                    assert fresh.last_kernel_name() == ("k_fused_t1" if depth == 12 else "k_fused2")
    finally:
        fresh.close()


def test_mixed_sequence_on_one_context(oracle):
    """One context, a sequence of batches that differ in everything the context keeps state about: sample
    type, size (full grid / small grid), bit depth, resampler, statistics (hint right / wrong), black frames
    (tier steering), batch length (frame groups, weighted rounds with clocks fed back between launches)."""
    import torch

    rng = np.random.default_rng(20261004)
    fresh = h.Context(0)

    def run(d, host, conv):
        dev_in = [[conv(p) for p in fr] for fr in host]
        dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
        torch.cuda.synchronize()
        fresh.convert_batch(d, dev_in, dev_out)
        od = _to_oracle_desc(d)
        for f, fr in enumerate(host):
            got = dev_out[f].cpu().numpy().view(np.uint16)
            want = oracle.convert_frame(od, fr)
            assert np.array_equal(got, want), f"{d.width}x{d.height} frame {f}: {np.count_nonzero(got != want)} samples differ"

    f32 = lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda()
    i16 = lambda p: torch.from_numpy(np.ascontiguousarray(p).view(np.int16)).cuda()

    def frames_f32(n, w, hh, top=1.0, black=()):
        out = []
        for k in range(n):
            planes = [rng.uniform(0.0, 1.0, w * hh).astype(np.float32) for _ in range(3)]
            for p in planes:
                p[k] = top
                if k in black:
                    p[:] = 0.0
            out.append(planes)
        return out

    try:
        big = (2048, 512)  # 131 072 tiles: the full grid of 256 blocks, XCD layout
        for step in range(2):
            run(h.make_desc(*big, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0), frames_f32(8, *big), f32)
            run(h.make_desc(256, 64, dst_depth=10, dst_matrix=h.MATRIX_BT709, resampler=1), frames_f32(5, 256, 64), f32)
            hf = [[p.astype(np.float16).view(np.uint16) for p in fr] for fr in frames_f32(4, 1024, 256)]
            run(h.make_desc(1024, 256, sample=h.SAMPLE_F16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0), hf, i16)
            u16 = [[rng.integers(0, 65536, 512 * 128, dtype=np.uint16) for _ in range(3)] for _ in range(6)]
            run(h.make_desc(512, 128, sample=h.SAMPLE_U16, src_depth=16, dst_depth=10, dst_matrix=h.MATRIX_BT2020NC, resampler=0,
                            src_transfer=16, dst_transfer=16), u16, i16)
            run(h.make_desc(*big, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0), frames_f32(8, *big, top=2.5), f32)  # hint wrong
            run(h.make_desc(*big, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0), frames_f32(4, *big, black=(0, 1, 2, 3)), f32)
            run(h.make_desc(*big, dst_depth=16, dst_matrix=h.MATRIX_YDZDX, chroma=h.CHROMA_444), frames_f32(2, *big), f32)
            run(h.make_desc(132, 36, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0), frames_f32(3, 132, 36), f32)
            run(h.make_desc(*big, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=1), frames_f32(3, *big), f32)
    finally:
        fresh.close()


def test_bench_device_generator_matches_host_generator():
    """bench.py builds its input on the device (closed form of the SURVEY 8c LCG in 64-bit integers): same bits as
    hdr2yuv_amd/synth.py, which tests/test_host_logic.py pins to the oracle's generator."""
    import importlib.util
    import torch

    from hdr2yuv_amd.synth import synth_frame

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(GOLD), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for (w, hh, f16) in ((256, 64, False), (1920, 1080, False), (512, 128, True)):
        gen = bench.DeviceSynth(3 * w * hh, torch.device("cuda", 0))
        for k in (0, 1, 63, 511):
            got = [p.cpu().numpy() for p in gen.frame(w, hh, k, f16)]
            want = synth_frame(w, hh, k, f16)
            for c in range(3):
                assert np.array_equal(got[c].view(np.uint16) if f16 else got[c].view(np.uint32), want[c].view(np.uint16) if f16 else want[c].view(np.uint32)), (w, hh, f16, k, c)


@pytest.mark.parametrize("w,hh", [(2, 2), (4, 2), (6, 10), (64, 32), (130, 18), (258, 70), (1920, 1080)])
def test_upsample_444(ctx, oracle, w, hh):
    """h2y_upsample_444 (Subsample420to444, convert.cpp:1869-1986) vs the oracle, itself pinned to the
    compiled reference function: replication and the FIR pair, tile edges and picture edges (sizes below one
    tile, one sample past a tile), full and video-range clamps, extreme code values."""
    import torch

    rng = np.random.default_rng(w * 7 + hh)
    for depth in (10, 16):
        maxcv = (1 << depth) - 1
        src = rng.integers(0, 1 << depth, (hh // 2, w // 2)).astype(np.uint16)
        src.flat[: min(4, src.size)] = [0, maxcv, maxcv, 0][: min(4, src.size)]
        dsrc = torch.from_numpy(src.view(np.int16)).cuda()
        for alg in (0, 1, 7):
            for (lo, hi) in ((0, maxcv), (16 << (depth - 8), 240 << (depth - 8))):
                ddst = torch.zeros(hh * w, dtype=torch.int16, device="cuda")
                torch.cuda.synchronize()  # the context's stream does not wait for torch's
                ctx.upsample_444(w, hh, alg, lo, hi, dsrc, ddst)
                got = ddst.cpu().numpy().view(np.uint16).reshape(hh, w)
                want = oracle.up444(src, w, hh, alg, lo, hi)
                assert np.array_equal(got, want), (w, hh, depth, alg, lo, hi, int(np.count_nonzero(got != want)))
    with pytest.raises(h.H2YError):
        ctx.upsample_444(w + 1, hh, 1, 0, 1023, dsrc, ddst)  # odd width: no defined bytes in the reference


def test_inverse_420_flow(ctx, oracle):
    """.yuv 4:2:0 -> G,B,R: both chroma planes upsampled (yuv2tiff.cpp:341-342), then matrix_inverse."""
    import torch

    rng = np.random.default_rng(420)
    for (w, hh) in ((256, 64), (68, 10)):
        n = w * hh
        for (mat, ind, outd, full, alg) in ((9, 12, 16, 0, 1), (1, 10, 10, 0, 1), (11, 12, 16, 0, 0), (1, 12, 12, 1, 1)):
            y = rng.integers(0, 1 << ind, n).astype(np.uint16)
            cb = rng.integers(0, 1 << ind, n // 4).astype(np.uint16)
            cr = rng.integers(0, 1 << ind, n // 4).astype(np.uint16)
            din = [torch.from_numpy(p.view(np.int16)).cuda() for p in (y, cb, cr)]
            dout = [torch.zeros(n, dtype=torch.int16, device="cuda") for _ in range(3)]
            torch.cuda.synchronize()  # the context's stream does not wait for torch's
            ctx.inverse_420(w, hh, ind, full, mat, outd, alg, din, dout)
            maxcv = (1 << ind) - 1
            full_planes = [y, oracle.up444(cb, w, hh, alg, 0, maxcv).reshape(-1), oracle.up444(cr, w, hh, alg, 0, maxcv).reshape(-1)]
            want = oracle.matrix_inverse(w, hh, ind, full, mat, outd, full_planes)
            for c in range(3):
                got = dout[c].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[c]), (w, hh, mat, ind, outd, full, alg, c, int(np.count_nonzero(got != want[c])))


FIRF_SIZES = [(4, 2), (8, 4), (240, 8), (244, 6), (252, 130), (480, 264), (484, 12), (1000, 300), (1920, 540)]


@pytest.mark.parametrize("w,hh", FIRF_SIZES)
def test_fir_fused_kernel_geometries(oracle, w, hh):
    """k_fir_fused (the FIR resampler in one pass: halo lanes, DPP taps, register history, integer FIR stages) vs the
    oracle: widths of one lane, one strip, one strip + one lane, several strips with a partial last one; heights below
    the filter's support, one segment, several segments (cuts recompute three row pairs either side); every matrix the
    first tier covers, 8/10/12 bits, both ranges, both normalisation variants, half input, pictures with black bars
    and out-of-table samples (every pixel of those takes the binary64 tier inside the loop)."""
    import torch

    rng = np.random.default_rng(w * 3 + hh)
    fresh = None
    try:
        for (mat, depth, full, sample, stats) in ((h.MATRIX_BT2020NC, 12, 0, h.SAMPLE_F32, None), (h.MATRIX_BT709, 10, 1, h.SAMPLE_F32, [(0, 1)] * 3),
                                                  (h.MATRIX_YDZDX, 12, 0, h.SAMPLE_F32, [(-1, 2)] * 3), (h.MATRIX_BT2020NC, 10, 0, h.SAMPLE_F16, [(0, 1)] * 3),
                                                  (h.MATRIX_BT709, 8, 0, h.SAMPLE_F32, None)):
            if fresh is not None:
                fresh.close()
            fresh = h.Context(0)  # tier steering starts from scratch: the first batch takes the first tier
            fresh.set_option("fir", "fused")
            d = h.make_desc(w, hh, sample=sample, dst_depth=depth, dst_matrix=mat, resampler=1, full_range=full, stats=stats)
            od = _to_oracle_desc(d)
            n = 3
            host = []
            for k in range(n):
                planes = _picture_like(rng, w, hh) if (k == 1 and w >= 48 and hh >= 16) else _rand_planes(rng, w, hh, h.SAMPLE_F32, plant=(w * hh >= 2))
                if sample == h.SAMPLE_F16:
                    planes = [np.asarray(p, np.float32).astype(np.float16).view(np.uint16) for p in planes]
                host.append(planes)
            conv = (lambda p: torch.from_numpy(np.ascontiguousarray(p).view(np.int16)).cuda()) if sample == h.SAMPLE_F16 else \
                   (lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda())
            dev_in = [[conv(p) for p in fr] for fr in host]
            for rnd in range(2):  # round 0: statistics from a pre-pass (normalising variant); round 1: the hint (identity variant)
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                fresh.convert_batch(d, dev_in, dev_out)
                if rnd == 0:  # (a batch dense in out-of-table samples sends the NEXT ones to the binary64 two-pass form)
                    assert fresh.last_kernel_name() == "k_fir_fused", fresh.last_kernel_variant()
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    want = oracle.convert_frame(od, host[f])
                    bad = np.flatnonzero(got != want)
                    assert bad.size == 0, f"{w}x{hh} mat {mat} depth {depth} round {rnd} frame {f}: {bad.size} samples differ, first at {bad[:8]} ({fresh.last_kernel_variant()})"
    finally:
        if fresh is not None:
            fresh.close()


@pytest.mark.parametrize("res", [0, 1])
def test_two_batches_in_flight(oracle, res):
    """h2y_convert_batch_enqueue twice before the first h2y_batch_finish: the second launch is queued behind the
    first; batches finish in order, each with its own statistics check (one frame of EACH batch breaks the hint and
    is redone while the other batch may still be running), a third enqueue is refused, and a stage entry in between
    is refused too."""
    import torch

    rng = np.random.default_rng(2002 + res)
    w, hh = 512, 96
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=res)
    od = _to_oracle_desc(d)
    batches = []
    for b in range(4):
        host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(8)]
        if b in (1, 2):
            host[b + 2][1][77] = np.float32(2.5)  # ceiling 2: this frame is redone
        dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
        dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
        batches.append((host, dev_in, dev_out))
    torch.cuda.synchronize()
    c = h.Context(0)
    if res:
        c.set_option("fir", "fused")
    try:
        c.convert_batch(d, batches[0][1], batches[0][2])  # establishes the hint
        c.convert_batch_enqueue(d, batches[1][1], batches[1][2])
        c.convert_batch_enqueue(d, batches[2][1], batches[2][2])
        with pytest.raises(h.H2YError):
            c.convert_batch_enqueue(d, batches[3][1], batches[3][2])
        with pytest.raises(h.H2YError):
            c.pic_stats(d, batches[3][1][0])
        assert c.batch_finish() == 1
        c.convert_batch_enqueue(d, batches[3][1], batches[3][2])  # a slot is free again
        assert c.batch_finish() == 1
        assert c.batch_finish() == 0
        assert c.batch_finish() == 0  # nothing in flight: a no-op
        for b, (host, _, dev_out) in enumerate(batches):
            for f in range(len(host)):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                want = oracle.convert_frame(od, host[f])
                assert np.array_equal(got, want), f"batch {b} frame {f}: {np.count_nonzero(got != want)} samples differ"
    finally:
        c.close()


def _cli(args):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hdr2yuv_amd", "hdr2yuv")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "hdr2yuv_amd", "cli"), "--no-print-directory"], check=True)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r


def test_cli_reads_raw_files(tmp_path, oracle):
    """The host program on real input files, every raw format it takes (hdr2yuv.cpp:582-656 for the integer ones):
    .f32 planar float (two frames, default resampler = FIR), .f16, .rgb 16-bit in R,G,B file order -> planes 2,0,1,
    .yuv 16-bit with --src_start_frame 1 --n_frames 3 out of five frames; appended bytes (tiff.cpp:440) vs the oracle."""
    rng = np.random.default_rng(808)
    w, hh = 128, 32
    n = w * hh

    # .f32: G,B,R planes per frame, LINEAR -> PQ, 10-bit BT.709 4:2:0, resampler left at its default (FIR)
    frames = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(2)]
    src, dst = tmp_path / "in.f32", tmp_path / "f32.yuv"
    src.write_bytes(b"".join(p.tobytes() for fr in frames for p in fr))
    _cli(["--src_filename", src, "--dst_filename", dst, "--src_pic_width", w, "--src_pic_height", hh, "--n_frames", 2, "--src_bit_depth", 32, "--dst_bit_depth", 10,
          "--src_transfer_characteristics", 8, "--dst_transfer_characteristics", 16, "--dst_matrix_coeffs", 1, "--dst_chroma_format_idc", 1,
          "--dst_video_full_range_flag", 0])
    od = ob.make_desc(w, hh, dst_depth=10, dst_matrix=1, resampler=1)
    want = np.concatenate([oracle.convert_frame(od, fr) for fr in frames])
    assert np.array_equal(np.fromfile(dst, np.uint16), want)

    # .f16: half planes, 12-bit BT.2020 box, full range
    hframes = [[p.astype(np.float16).view(np.uint16) for p in frames[0]]]
    src, dst = tmp_path / "in.f16", tmp_path / "f16.yuv"
    src.write_bytes(b"".join(p.tobytes() for fr in hframes for p in fr))
    _cli(["--src_filename", src, "--dst_filename", dst, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 16, "--dst_bit_depth", 12,
          "--src_transfer_characteristics", 8, "--dst_transfer_characteristics", 16, "--dst_matrix_coeffs", 9, "--dst_chroma_format_idc", 1,
          "--dst_video_full_range_flag", 1, "--chroma_resampler_type", 0])
    od = ob.make_desc(w, hh, sample=ob.SAMPLE_F16, dst_depth=12, dst_matrix=9, resampler=0, full_range=1)
    assert np.array_equal(np.fromfile(dst, np.uint16), oracle.convert_frame(od, hframes[0]))

    # .rgb: 16-bit samples, file order R,G,B -> memory planes 2,0,1; PQ in, PQ out (no transfer change), 16 -> 10 bits
    r, g, b = [rng.integers(0, 65536, n).astype(np.uint16) for _ in range(3)]
    src, dst = tmp_path / "in.rgb", tmp_path / "rgb.yuv"
    src.write_bytes(r.tobytes() + g.tobytes() + b.tobytes())
    _cli(["--src_filename", src, "--dst_filename", dst, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 16, "--dst_bit_depth", 10,
          "--src_transfer_characteristics", 16, "--dst_transfer_characteristics", 16, "--dst_matrix_coeffs", 9, "--src_chroma_format_idc", 3,
          "--dst_chroma_format_idc", 1, "--src_video_full_range_flag", 0, "--chroma_resampler_type", 1])
    od = ob.make_desc(w, hh, sample=ob.SAMPLE_U16, src_depth=16, dst_depth=10, src_transfer=16, dst_transfer=16, dst_matrix=9, resampler=1)
    assert np.array_equal(np.fromfile(dst, np.uint16), oracle.convert_frame(od, [g, b, r]))

    # .yuv: five 16-bit 4:4:4 frames in the file, frames 1..3 converted (Y'DzDx 4:4:4 out, 16 -> 12 bits), appended to an existing file
    yframes = [[rng.integers(0, 65536, n).astype(np.uint16) for _ in range(3)] for _ in range(5)]
    src, dst = tmp_path / "in.yuv", tmp_path / "yuv.yuv"
    src.write_bytes(b"".join(p.tobytes() for fr in yframes for p in fr))
    dst.write_bytes(b"\x01\x02" * 8)  # what is already there stays
    _cli(["--src_filename", src, "--dst_filename", dst, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 16, "--dst_bit_depth", 12,
          "--src_start_frame", 1, "--n_frames", 3, "--src_transfer_characteristics", 16, "--dst_transfer_characteristics", 16,
          "--dst_matrix_coeffs", 11, "--src_chroma_format_idc", 3, "--dst_chroma_format_idc", 3, "--src_video_full_range_flag", 0])
    od = ob.make_desc(w, hh, sample=ob.SAMPLE_U16, src_depth=16, dst_depth=12, src_transfer=16, dst_transfer=16, dst_matrix=11, chroma=3)
    got = dst.read_bytes()
    assert got[:16] == b"\x01\x02" * 8
    want = np.concatenate([oracle.convert_frame(od, yframes[k]) for k in (1, 2, 3)])
    assert np.array_equal(np.frombuffer(got[16:], np.uint16), want)


@pytest.mark.parametrize("path", ["t1_box", "t2_box", "fir_fused"])
def test_dark_frames_dense_below_the_tables(oracle, path):
    """Whole frames of dark samples -- uniform noise to the fourth power: 1.6 % of the samples below 2^-24, every wave meets
    several per tile row -- through the kernels that answer them in line from the full-range table by scalar loads
    (pq_ext_inline): k_fused_t1's redo passes, k_fused2's sample step, k_fir_fused's binary64 branch."""
    import torch

    rng = np.random.default_rng(4444)
    w, hh, n = 512, 128, 6
    host = []
    for k in range(n):
        planes = [(rng.random(w * hh, dtype=np.float32) ** 4).astype(np.float32) for _ in range(3)]
        for p in planes:
            p[0], p[1] = 0.0, 1.0
        host.append(planes)
    kw = dict(dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=1 if path == "fir_fused" else 0)
    d = h.make_desc(w, hh, **kw)
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    c = h.Context(0)
    try:
        if path == "t2_box":
            c.set_option("t1", "0")
        if path == "fir_fused":
            c.set_option("fir", "fused")
        dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
        for rnd in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            if rnd == 0:
                assert c.last_kernel_name() == {"t1_box": "k_fused_t1", "t2_box": "k_fused2", "fir_fused": "k_fir_fused"}[path], c.last_kernel_variant()
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), (path, rnd, f, int(np.count_nonzero(got != want[f])))
    finally:
        c.close()


@pytest.mark.parametrize("content", ["uniform", "bars", "dark", "low"])
@pytest.mark.parametrize("out", ["box", "444", "fir_twopass"])
def test_dynamic_last_frame(oracle, content, out):
    """k_fused_t1 with the last frame of every frame group drawn dynamically from the counters in global memory (h2y_walk.h,
    "The dynamic last frame"; "tail" = "on": two frames per group suffice): batches of 12 and 5 frames at one, two and four
    groups, second round on the statistics hint; uniform noise, letterbox bars (rows of zeros: every tile of them joins the
    redo lists in that frame), dark frames (samples below the tables) and a frame with a sample <= -1 in its LAST frame (the
    floor assumed for the batch is wrong there: it must be found from the tail's own minimum and the frame redone)."""
    import torch

    rng = np.random.default_rng({"uniform": 1, "bars": 2, "dark": 3, "low": 4}[content] * 97 + len(out))
    w, hh = 1024, 256  # 32 chunks a frame: batches of five frames and more fill the 256-block grid (the XCD layout needs whole rounds)
    kw = dict(dst_depth=12, dst_matrix=h.MATRIX_BT2020NC)
    if out == "444":
        kw.update(chroma=h.CHROMA_444, dst_matrix=h.MATRIX_YDZDX, dst_depth=10)
    elif out == "fir_twopass":
        kw.update(resampler=1, dst_depth=10)
    else:
        kw.update(resampler=0)
    d = h.make_desc(w, hh, **kw)
    od = _to_oracle_desc(d)
    for n, groups in ((12, 2), (12, 4), (5, 1), (16, 8)):
        host = []
        for k in range(n):
            planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
            if content == "bars":
                for p in planes:
                    p[: 12 * w] = 0.0
                    p[-12 * w:] = 0.0
                    p[12 * w], p[12 * w + 1] = 0.0, 1.0
            elif content == "dark":  # squared: 0.02 % of the samples below the tables (denser, and the context moves to k_fused2)
                planes = [(p * p).astype(np.float32) for p in planes]
                for p in planes:
                    p[0], p[1] = 0.0, 1.0
            elif content == "low" and k >= n - groups:
                planes[1][w * 40 + 7] = np.float32(-1.5)  # in the dynamic frames only
            host.append(planes)
        want = [oracle.convert_frame(od, fr) for fr in host]
        c = h.Context(0)
        try:
            c.set_option("tail", "on")
            c.set_option("groups", str(groups))
            if out == "fir_twopass":
                c.set_option("fir", "twopass")
            dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
            for rnd in range(3):
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                c.convert_batch(d, dev_in, dev_out)
                assert c.last_kernel_name() == "k_fused_t1", c.last_kernel_variant()
                assert " tail=1" in c.last_kernel_variant(), c.last_kernel_variant()
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    assert np.array_equal(got, want[f]), (content, out, n, groups, rnd, f, int(np.count_nonzero(got != want[f])))
        finally:
            c.close()


def test_inverse_frame_argument_errors(ctx):
    """h2y_inverse_frame / h2y_inverse_420 refuse what the reference cannot do either: matrix_coeffs 0 (convert.cpp:1733-1736
    exits), chroma formats other than 4:4:4 / 4:2:0, odd 4:2:0 sizes (Subsample420to444 reads rows it never wrote), depths
    outside 8..16; the context stays usable."""
    n = 64 * 16
    planes = [np.zeros(n, np.uint16) for _ in range(3)]
    for args, code in (((64, 16, 3, 12, 0, 0, 16, 0), 2), ((64, 16, 2, 12, 0, 1, 16, 0), 2), ((66, 16, 1, 12, 0, 1, 16, 1), 1),
                       ((64, 16, 3, 7, 0, 1, 16, 0), 1), ((64, 16, 1, 12, 0, 1, 17, 1), 1)):
        with pytest.raises(h.H2YError) as e:
            ctx.inverse_frame(*args, planes)
        assert e.value.code == code, (args, str(e.value))
    out = ctx.inverse_frame(64, 16, 3, 12, 0, 11, 16, 0, planes)  # still works: Y'DzDx of zeros
    assert len(out) == 3 and out[0].shape == (n,)


def test_cli_runs_the_reference_test_sh_lines(tmp_path, oracle):
    """test.sh:21-57 and :66-74 flag for flag (tests/cli_lines.py): no range flag anywhere, so the reference's destination is
    VIDEO range (in_pic zeroed at hdr2yuv.cpp:765, copied to the destination at :296-297) -- also on the .exr line, where
    read_exr() forces only the input picture to full range (exr.cpp:183).  Synthetic files of the lines' own sizes."""
    from cli_lines import TEST_SH

    rng = np.random.default_rng(2121)
    w, hh = 2560, 1600
    n = w * hh
    planes = [rng.integers(0, 4096, n).astype(np.uint16) for _ in range(3)]
    (tmp_path / "in.yuv").write_bytes(b"".join(p.tobytes() for p in planes))  # file order Y, Cb, Cr = planes 0,1,2
    (tmp_path / "in.rgb").write_bytes(b"".join(p.tobytes() for p in planes))  # file order R, G, B = planes 2,0,1
    for name, mem, od in (
        ("yuv444_to_444", planes, ob.make_desc(w, hh, sample=ob.SAMPLE_U16, src_depth=12, dst_depth=12, src_transfer=1, dst_transfer=1, src_matrix=1,
                                               dst_matrix=1, src_primaries=1, dst_primaries=1, full_range=0, chroma=3, resampler=1)),
        ("yuv444_to_420", planes, ob.make_desc(w, hh, sample=ob.SAMPLE_U16, src_depth=12, dst_depth=12, src_transfer=1, dst_transfer=1, src_matrix=1,
                                               dst_matrix=1, src_primaries=1, dst_primaries=1, full_range=0, chroma=1, resampler=1)),
        ("rgb_to_420_10b", [planes[1], planes[2], planes[0]],
         ob.make_desc(w, hh, sample=ob.SAMPLE_U16, src_depth=12, dst_depth=10, src_transfer=1, dst_transfer=1, src_matrix=0, dst_matrix=1,
                      src_primaries=1, dst_primaries=1, full_range=0, chroma=1, resampler=1)),
    ):
        dst = tmp_path / (name + ".yuv")
        _cli(TEST_SH[name].format(src=tmp_path / "in", dst=tmp_path / name).split())
        assert np.array_equal(np.fromfile(dst, np.uint16), oracle.convert_frame(od, mem)), name
    # the .exr line: half planes as read_exr() leaves them; LINEAR -> BT.709 (bt1886_r), 10-bit BT.709 4:2:0, video range
    w, hh = 1920, 1080
    half = [p.astype(np.float16).view(np.uint16) for p in _rand_planes(rng, w, hh, h.SAMPLE_F32)]
    (tmp_path / "in.f16").write_bytes(b"".join(p.tobytes() for p in half))
    _cli(TEST_SH["exr_to_420_10b"].format(src=tmp_path / "in", dst=tmp_path / "exr").split())
    od = ob.make_desc(w, hh, sample=ob.SAMPLE_F16, dst_depth=10, src_transfer=8, dst_transfer=1, src_matrix=0, dst_matrix=1,
                      src_primaries=1, dst_primaries=1, full_range=0, chroma=1, resampler=1)
    assert np.array_equal(np.fromfile(tmp_path / "exr.yuv", np.uint16), oracle.convert_frame(od, half))


def test_cli_default_range_is_the_reference_s(tmp_path, oracle):
    """A float input with no range flag at all: video range out (235 x D scale, [16 D, 235 D] / [16 D, 240 D] clamps);
    --src_video_full_range_flag 1 alone makes the destination full range (copied at parse time); an explicit
    --dst_video_full_range_flag wins over both."""
    rng = np.random.default_rng(99)
    w, hh = 128, 32
    frame = _rand_planes(rng, w, hh, h.SAMPLE_F32)
    src = tmp_path / "in.f32"
    src.write_bytes(b"".join(p.tobytes() for p in frame))
    common = ["--src_filename", src, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 32, "--dst_bit_depth", 10,
              "--src_transfer_characteristics", 8, "--dst_transfer_characteristics", 16, "--dst_matrix_coeffs", 9, "--dst_chroma_format_idc", 1]
    for extra, full in (([], 0), (["--src_video_full_range_flag", 1], 1), (["--src_video_full_range_flag", 1, "--dst_video_full_range_flag", 0], 0)):
        dst = tmp_path / f"o{len(extra)}.yuv"
        _cli(common + ["--dst_filename", dst] + extra)
        od = ob.make_desc(w, hh, dst_depth=10, dst_matrix=9, src_primaries=0, dst_primaries=0, resampler=1, full_range=full)
        assert np.array_equal(np.fromfile(dst, np.uint16), oracle.convert_frame(od, frame)), extra


def test_cli_frame_blocks_on_two_contexts(tmp_path, oracle):
    """--gpus 2 (both contexts on device 0 here: --devices 0,0): two host threads, each with its own context and pinned
    ring, frames 0..3 and 4..6 of seven; every frame lands at `old size + k x frame bytes`, so the file is what seven
    appending runs would have left (tiff.cpp:440) and what the single-context run writes."""
    rng = np.random.default_rng(4242)
    w, hh, nf = 192, 48, 9
    frames = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(nf)]
    src = tmp_path / "in.f32"
    src.write_bytes(b"".join(p.tobytes() for fr in frames for p in fr))
    args = ["--src_filename", src, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 32, "--dst_bit_depth", 12,
            "--src_transfer_characteristics", 8, "--dst_transfer_characteristics", 16, "--dst_matrix_coeffs", 9, "--dst_chroma_format_idc", 1,
            "--src_start_frame", 1, "--n_frames", 7, "--chroma_resampler_type", 1]
    one, two = tmp_path / "one.yuv", tmp_path / "two.yuv"
    for path in (one, two):
        path.write_bytes(b"\x07" * 10)  # what is already in the file stays in front
    _cli(args + ["--dst_filename", one])
    r = _cli(args + ["--dst_filename", two, "--gpus", 2, "--devices", "0,0", "--verbose_level", 1])
    assert "frame 3:" in r.stdout and "frame 6:" in r.stdout
    od = ob.make_desc(w, hh, dst_depth=12, dst_matrix=9, src_primaries=0, dst_primaries=0, resampler=1, full_range=0)
    want = b"\x07" * 10 + np.concatenate([oracle.convert_frame(od, frames[k]) for k in range(1, 8)]).tobytes()
    assert one.read_bytes() == want
    assert two.read_bytes() == want


def test_cli_inverse_flow_writes_planar_rgb(tmp_path, oracle, ctx):
    """.yuv in, .rgb out: matrix_inverse() (hdr2yuv.cpp:818-819) with write_tiff()'s shift (tiff.cpp:564), the samples it
    would interleave written as planes R, G, B; test.sh:78-86's flags (12-bit BT.709 4:4:4 -> 16 bits), and a 4:2:0 file
    through the upsampler first (yuv2tiff.cpp:341-342).  h2y_inverse_frame is the host-buffer entry underneath."""
    rng = np.random.default_rng(8686)
    w, hh = 256, 64
    n = w * hh
    y, cb, cr = [rng.integers(0, 4096, n).astype(np.uint16) for _ in range(3)]
    src, dst = tmp_path / "in.yuv", tmp_path / "out.rgb"
    src.write_bytes(y.tobytes() + cb.tobytes() + cr.tobytes())
    _cli(("--src_matrix_coeffs 1 --dst_matrix_coeffs 0 --src_transfer_characteristics 1 --dst_transfer_characteristics 1 "
          f"--src_colour_primaries 1 --dst_colour_primaries 1 --src_filename {src} --dst_filename {dst} "
          f"--src_pic_width {w} --src_pic_height {hh} --src_bit_depth 12 --dst_bit_depth 16 "
          "--src_chroma_format_idc 3 --dst_chroma_format_idc 3 --verbose_level 4 --src_start_frame 0").split())
    g, b, r = oracle.matrix_inverse(w, hh, 12, 0, 1, 16, [y, cb, cr])
    assert np.array_equal(np.fromfile(dst, np.uint16), np.concatenate([r, g, b]))
    got = ctx.inverse_frame(w, hh, 3, 12, 0, 1, 16, 0, [y, cb, cr])
    assert all(np.array_equal(a, b_) for a, b_ in zip(got, (g, b, r)))
    # 4:2:0 input, Y'DzDx, FIR upsampler
    cb2, cr2 = cb[: n // 4], cr[: n // 4]
    src.write_bytes(y.tobytes() + cb2.tobytes() + cr2.tobytes())
    dst = tmp_path / "out2.rgb"
    _cli(["--src_filename", src, "--dst_filename", dst, "--src_pic_width", w, "--src_pic_height", hh, "--src_bit_depth", 12, "--dst_bit_depth", 16,
          "--src_matrix_coeffs", 11, "--src_chroma_format_idc", 1, "--dst_chroma_format_idc", 3, "--chroma_resampler_type", 1])
    full = [y, oracle.up444(cb2, w, hh, 1, 0, 4095).reshape(-1), oracle.up444(cr2, w, hh, 1, 0, 4095).reshape(-1)]
    g, b, r = oracle.matrix_inverse(w, hh, 12, 0, 11, 16, full)
    assert np.array_equal(np.fromfile(dst, np.uint16), np.concatenate([r, g, b]))


def test_context_options_and_variant_names(oracle):
    """h2y_ctx_set_option: every knob is per context (the library reads nothing from the environment), none changes a
    byte; h2y_last_kernel_variant names the instantiation that ran, so a test can say which kernel it covered whatever ran
    before it."""
    import torch

    rng = np.random.default_rng(77)
    w, hh = 512, 128
    host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(8)]
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0, stats=[(0, 1)] * 3)
    want = [oracle.convert_frame(_to_oracle_desc(d), fr) for fr in host]
    expect = {(): "k_fused_t1<F32,420BOX,YCBCR,PQ_IDENT> groups=8", (("t1", "0"),): "k_fused2<F32,420BOX,YCBCR,PQ_IDENT> groups=8",
              (("groups", "2"),): "k_fused_t1<F32,420BOX,YCBCR,PQ_IDENT> groups=2", (("groups", "1"), ("balance", "off")): "k_fused_t1<F32,420BOX,YCBCR,PQ_IDENT> groups=1"}
    for opts, name in expect.items():
        c = h.Context(0)
        try:
            for k, v in opts:
                c.set_option(k, v)
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            assert c.last_kernel_variant().startswith(name), (opts, c.last_kernel_variant())
            for f in range(len(host)):
                assert np.array_equal(dev_out[f].cpu().numpy().view(np.uint16), want[f]), (opts, f)
            with pytest.raises(h.H2YError):
                c.set_option("no_such_knob", "1")
            with pytest.raises(h.H2YError):
                c.set_option("balance", "0xFF,1.2")  # every XCD fast is no split at all
        finally:
            c.close()


@pytest.mark.parametrize("path", ["t1_box", "t2_box", "fir_fused", "fir_twopass", "frame_entry", "ydzdx16_444"])
def test_samples_below_the_tables(oracle, path):
    """Positive samples below 2^-24 (pictures that never went through half floats hold them): outside the LDS tables of both
    fast tiers, answered by pq_slow() from the table in global memory (pq_build_table_ext) unless ambiguous or subnormal --
    2 % of the samples here, over the whole exponent range, plus subnormals, negative tiny values and exact zeros, through
    every kernel that can meet them."""
    import torch

    rng = np.random.default_rng(2424)
    w, hh, n = 256, 64, 5
    host = []
    for k in range(n):
        planes = _rand_planes(rng, w, hh, h.SAMPLE_F32)
        for p in planes:
            idx = rng.choice(p.size - 2, p.size // 50, replace=False) + 2
            p[idx] = (10.0 ** rng.uniform(-37.9, -7.3, idx.size)).astype(np.float32)
            sub = rng.choice(p.size - 2, 12, replace=False) + 2
            p[sub[:4]] = np.float32(1e-40)    # subnormal
            p[sub[4:8]] = np.float32(-1e-12)  # negative: pow() of it is NaN in the reference
            p[sub[8:]] = 0.0
        host.append(planes)
    kw = dict(dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0)
    opts = {}
    if path == "t2_box":
        opts["t1"] = "0"
    elif path == "fir_fused":
        kw.update(dst_depth=10, resampler=1)
        opts["fir"] = "fused"
    elif path == "fir_twopass":
        kw.update(dst_depth=10, resampler=1)
        opts["fir"] = "twopass"
    elif path == "ydzdx16_444":
        kw = dict(dst_depth=16, dst_matrix=h.MATRIX_YDZDX, chroma=h.CHROMA_444)
    d = h.make_desc(w, hh, **kw)
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    c = h.Context(0)
    try:
        for name, value in opts.items():
            c.set_option(name, value)
        if path == "frame_entry":
            for f in range(n):
                assert np.array_equal(c.convert_frame(d, host[f]), want[f]), f
            return
        dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
        for rnd in range(2):  # round 1 runs on round 0's statistics hint
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            if rnd == 0:
                kernel, variant = c.last_kernel_name(), c.last_kernel_variant()
                assert kernel == {"t1_box": "k_fused_t1", "t2_box": "k_fused2", "fir_fused": "k_fir_fused", "fir_twopass": "k_fused_t1",
                                  "ydzdx16_444": "k_fused2"}[path], variant
                assert ("+k_fir420" in variant) == (path == "fir_twopass"), variant
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"{path} round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
    finally:
        c.close()


@pytest.mark.parametrize("path,full", [("t1_box", 0), ("t1_box", 1), ("fir_fused", 0), ("fir_fused", 1), ("ydzdx_box", 0)])
def test_first_tier_sentinel_segments_and_chroma_range(oracle, path, full):
    """The first tier (pq_t1, h2y_math.h) answers from a table whose records are sentinels (value NaN: the pixel goes to the
    binary64 tier) for inputs outside [2^-24, 1 + 2^-8) AND for the few segments in which PQ passes a power of two; and it
    gives the chroma raw, the offset and the clamp to maxCV left out because t1_chroma_in_range() proved per launch that the
    clamp cannot act on a pixel it settles.  Samples packed where that could go wrong: exponents uniform over [2^-26, 4)
    (every binade of the table and both ends), dense runs of consecutive floats across the inputs whose PQ value is
    2^-7 .. 2^0 (the sentinel segments and their neighbours), values in [1, 2), pixels with one plane near 0 and the others
    near 1 and the reverse (the chroma's extremes: at full range the raw integer reaches -(Half - 1) and maxCV - (Half - 1)),
    exact 1.0 and exact 0.0.  Box and one-pass FIR kernels, video and full range, YCbCr and YDzDx."""
    import torch

    rng = np.random.default_rng(77 + full)
    w, hh, n = 256, 64, 4

    def pq_inv(v):  # ST 2084 EOTF, double precision: where the table's output passes v
        m1, m2, c1, c2, c3 = 0.1593017578125, 78.84375, 0.8359375, 18.8515625, 18.6875
        p = v ** (1.0 / m2)
        return (max(p - c1, 0.0) / (c2 - c3 * p)) ** (1.0 / m1)

    host = []
    for k in range(n):
        planes = [np.exp2(rng.uniform(-26.0, 2.0, hh * w)).astype(np.float32) for _ in range(3)]
        for flat in planes:
            pos = 0
            for e in range(0, 8):  # runs of consecutive floats around PQ^-1(2^-e)
                x0 = np.float32(min(pq_inv(2.0 ** -e), 1.0))
                b0 = int(x0.view(np.uint32)) - 700
                run = (np.arange(1400, dtype=np.uint32) + np.uint32(b0)).view(np.float32)
                flat[pos:pos + run.size] = run
                pos += run.size + 37
            flat[pos:pos + 300] = rng.uniform(1.0, 2.0, 300).astype(np.float32)
            flat[pos + 300:pos + 340] = 1.0
            flat[pos + 340:pos + 380] = 0.0
        # the chroma's extremes, row 40 on: (G, B, R) = (hi, lo, hi), (lo, hi, lo), (hi, hi, lo), (lo, lo, hi) with lo in [2^-24, 2^-20], hi in [0.98, 1.0039)
        lo = lambda m: np.exp2(rng.uniform(-24.0, -20.0, m)).astype(np.float32)
        hi = lambda m: rng.uniform(0.98, 1.0039, m).astype(np.float32)
        for j, pat in enumerate(((1, 0, 1), (0, 1, 0), (1, 1, 0), (0, 0, 1))):
            for c in range(3):
                planes[c][(40 + 2 * j) * w:(42 + 2 * j) * w] = (hi if pat[c] else lo)(2 * w)
        host.append(planes)
    kw = dict(dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=0, full_range=full)
    opts = {}
    if path == "fir_fused":
        kw.update(dst_depth=10, resampler=1)
        opts["fir"] = "fused"
    elif path == "ydzdx_box":
        kw.update(dst_matrix=h.MATRIX_YDZDX)
    d = h.make_desc(w, hh, **kw)
    od = _to_oracle_desc(d)
    want = [oracle.convert_frame(od, fr) for fr in host]
    c = h.Context(0)
    try:
        for name, value in opts.items():
            c.set_option(name, value)
        dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
        for rnd in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            if rnd == 0:
                assert c.last_kernel_name() == ("k_fir_fused" if path == "fir_fused" else "k_fused_t1"), c.last_kernel_variant()
            for f in range(n):
                got = dev_out[f].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want[f]), f"{path} full={full} round {rnd} frame {f}: {np.count_nonzero(got != want[f])} samples differ"
    finally:
        c.close()


@pytest.mark.parametrize("firsync", ["0", "1", "2", "8", "auto"])
def test_fir_fused_waves_in_step(oracle, firsync):
    """k_fir_fused keeps the sixteen waves of a block in step with a barrier every few steps (`firsync`).  The waves do not
    depend on each other, so the bytes cannot change; what must hold is that nothing hangs when the waves of a block have
    units of different length or none at all: widths whose strips do not fill a block (8, 3, 17 strips: a block then holds
    units of two or more frames / segments), rows cut unevenly by fixed XCD weights, more waves than units."""
    import torch

    rng = np.random.default_rng(31)
    c = h.Context(0)
    try:
        c.set_option("fir", "fused")
        c.set_option("firsync", firsync)
        c.set_option("balance", "0x55,1.2")
        for w, hh, n in ((1920, 270, 3), (720, 130, 5), (4080, 66, 2), (3840, 128, 1)):
            host = [_rand_planes(rng, w, hh, h.SAMPLE_F32) for _ in range(n)]
            d = h.make_desc(w, hh, dst_depth=10, dst_matrix=h.MATRIX_BT709, resampler=1)
            od = _to_oracle_desc(d)
            dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in host]
            for rnd in range(2):
                dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
                torch.cuda.synchronize()
                c.convert_batch(d, dev_in, dev_out)
                assert c.last_kernel_name() == "k_fir_fused", c.last_kernel_variant()
                for f in range(n):
                    got = dev_out[f].cpu().numpy().view(np.uint16)
                    want = oracle.convert_frame(od, host[f])
                    assert np.array_equal(got, want), (w, hh, rnd, f, int(np.count_nonzero(got != want)))
        with pytest.raises(h.H2YError):
            c.set_option("firsync", "-3")
    finally:
        c.close()


@pytest.mark.parametrize("balance", ["0x55,1.12", "0xAA,1.25", "0x0F,1.2"])
def test_fir_fused_rows_by_xcd_speed(oracle, balance):
    """k_fir_fused cuts every (frame, strip) column into its segments in proportion to the speeds of the XCDs its units run
    on (a launch ends with its slowest wave, and the XCDs differ).  Here the weights are fixed through the "balance" option,
    strong ones included, on a batch that fills the grid (16 x 4K: 4096 units): segments of unequal length, every row of
    every frame still produced exactly once -- frame 0 is the SURVEY 8c known answer, two more go through the oracle."""
    import torch

    case = KNOWN["cases"]["C2_4k_2020_12b_fir"]
    d = h.make_desc(**case["desc"])
    n = 16
    host = {k: oracle.synth_frame(d.width, d.height, k) for k in (0, 5, 11)}
    gen = None
    dev_in = []
    for k in range(n):
        planes = host[k] if k in host else oracle.synth_frame(d.width, d.height, k)
        dev_in.append([torch.from_numpy(p).cuda() for p in planes])
    c = h.Context(0)
    c.set_option("fir", "fused")
    c.set_option("balance", balance)
    try:
        for rnd in range(2):
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            c.convert_batch(d, dev_in, dev_out)
            assert c.last_kernel_name() == "k_fir_fused", c.last_kernel_variant()
            assert _md5(dev_out[0].cpu().numpy().view(np.uint16)) == case["md5"]
            od = _to_oracle_desc(d)
            for k in (5, 11):
                got = dev_out[k].cpu().numpy().view(np.uint16)
                want = oracle.convert_frame(od, host[k])
                assert np.array_equal(got, want), f"{balance} round {rnd} frame {k}: {np.count_nonzero(got != want)} samples differ"
    finally:
        c.close()


def test_fuzz_round2_paths_against_oracle(oracle):
    """Seeded sweep over what round 2 added: the FIR path forced onto k_fir_fused (any width % 4 == 0, any even height,
    1..12 frames, float or half input, 8..12 bits, both matrices it takes, measured / overridden statistics, fixed XCD
    weights), every transfer pair through the table tiers (with out-of-table and negative samples), and batches enqueued two
    at a time with different descriptors behind each other."""
    import torch

    rng = np.random.default_rng(20261005)
    tf_codes = [8, 16, 18, 1, 14]
    c = h.Context(0)
    c.set_option("fir", "fused")
    pending = []  # (desc, host frames, device outputs) of batches in flight

    def check(item):
        d, host, dev_out, tag = item
        od = _to_oracle_desc(d)
        for f, fr in enumerate(host):
            got = dev_out[f].cpu().numpy().view(np.uint16)
            want = oracle.convert_frame(od, fr)
            bad = np.flatnonzero(got != want)
            assert bad.size == 0, (tag, f, int(bad.size), bad[:6].tolist(), c.last_kernel_variant())

    try:
        for it in range(120):
            kind = it % 3
            sample = h.SAMPLE_F32 if rng.random() < 0.75 else h.SAMPLE_F16
            if kind == 0:  # fused FIR geometries
                w, hh = int(rng.integers(1, 130)) * 4, int(rng.integers(1, 90)) * 2
                kw = dict(dst_depth=int(rng.choice([8, 10, 12])), dst_matrix=int(rng.choice([h.MATRIX_BT2020NC, h.MATRIX_BT709, h.MATRIX_YDZDX])),
                          resampler=1, full_range=int(rng.integers(0, 2)), sample=sample)
                n = int(rng.integers(1, 13))
            elif kind == 1:  # transfer pairs, any output form
                w, hh = int(rng.integers(1, 40)) * 4, int(rng.integers(1, 30)) * 4
                src, dst = int(rng.choice(tf_codes)), int(rng.choice(tf_codes))
                kw = dict(dst_depth=int(rng.choice([10, 12, 16])), dst_matrix=int(rng.choice([h.MATRIX_BT2020NC, h.MATRIX_YDZDX, h.MATRIX_Y100])),
                          src_transfer=src, dst_transfer=dst, chroma=int(rng.choice([h.CHROMA_420, h.CHROMA_444])), resampler=int(rng.integers(0, 2)),
                          sample=h.SAMPLE_F32)
                sample = h.SAMPLE_F32
                n = int(rng.integers(1, 4))
            else:  # box / 4:4:4 batches with fixed XCD weights now and then
                w, hh = int(rng.integers(8, 200)) * 4, int(rng.integers(2, 60)) * 4
                kw = dict(dst_depth=int(rng.choice([10, 12, 16])), dst_matrix=int(rng.choice([h.MATRIX_BT2020NC, h.MATRIX_YDZDX])),
                          chroma=int(rng.choice([h.CHROMA_420, h.CHROMA_444])), resampler=0, sample=sample)
                n = int(rng.choice([1, 2, 8, 16]))
            mode = int(rng.integers(0, 3))
            if mode == 1:
                kw["stats"] = [(0, 1)] * 3
            if mode == 2 and kind != 1:
                kw["stats"] = [(-1, 2)] * 3
            d = h.make_desc(w, hh, **kw)
            host = []
            for _ in range(n):
                planes = _rand_planes(rng, w, hh, h.SAMPLE_F32, plant=(w * hh >= 2))
                if rng.random() < 0.3 and w * hh >= 64:
                    planes[1][7:19] = 0.0
                    planes[2][23] = np.float32(-0.25)
                    planes[0][31] = np.float32(1.5)
                if sample == h.SAMPLE_F16:
                    planes = [p.astype(np.float16).view(np.uint16) for p in planes]
                host.append(planes)
            conv = (lambda p: torch.from_numpy(np.ascontiguousarray(p).view(np.int16)).cuda()) if sample == h.SAMPLE_F16 else \
                   (lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda())
            dev_in = [[conv(p) for p in fr] for fr in host]
            dev_out = [torch.zeros(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in host]
            torch.cuda.synchronize()
            if it % 10 == 5 and not pending:
                c.set_option("balance", ["0x55,1.2", "0xF0,1.1", "off", "adaptive"][int(rng.integers(0, 4))])
            c.convert_batch_enqueue(d, dev_in, dev_out)
            pending.append((d, host, dev_out, (it, w, hh, n, kw), dev_in))
            if len(pending) == 2 or rng.random() < 0.4:  # finish the older one (sometimes right away, sometimes with one behind it)
                c.batch_finish()
                check(pending.pop(0)[:4])
        while pending:
            c.batch_finish()
            check(pending.pop(0)[:4])
    finally:
        c.close()
