"""Host-side logic: synthetic frames, sharding, and the two-rank CPU rehearsal of
the multi-GPU path (gloo)."""
import os
import socket
import sys

import numpy as np
import pytest

from hdr2yuv_amd.shard import frame_offset_bytes, frames_for_rank
from hdr2yuv_amd.synth import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_matches_oracle_generator(oracle):
    for f16 in (False, True):
        for frame in (0, 5):
            a = synth_frame(64, 32, frame, f16)
            b = oracle.synth_frame(64, 32, frame, f16)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
    p = synth_frame(16, 4)[0]
    assert p[0] == 0.0 and p[1] == 1.0 and p.min() >= 0 and p.max() <= 1.0


@pytest.mark.parametrize("n,world", [(512, 8), (512, 1), (10, 4), (3, 8), (0, 2), (7, 7)])
def test_frames_for_rank_partition(n, world):
    seen = []
    for r in range(world):
        seen += list(frames_for_rank(n, r, world))
    assert seen == list(range(n))
    sizes = [len(frames_for_rank(n, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1
    assert frame_offset_bytes(3, 100) == 300


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_frames, tmpdir):
    """What a GPU rank does, with the oracle standing in for the device: convert
    the frames this rank owns, write them at their offsets, reduce the counters."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from hdr2yuv_amd.shard import frame_offset_bytes as fob, frames_for_rank as ffr, reduce_counters
    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = ob.Oracle(build=False)
    w, h = 64, 32
    d = ob.make_desc(w, h, dst_depth=10, dst_matrix=ob.MATRIX_BT2020NC, resampler=1)
    fb = ob.frame_samples(d) * 2
    path = os.path.join(tmpdir, "out.yuv")
    if rank == 0:
        open(path, "wb").truncate(n_frames * fb)
    dist.barrier()
    px = 0
    with open(path, "r+b") as f:
        for k in ffr(n_frames, rank, world):
            out = o.convert_frame(d, o.synth_frame(w, h, k))
            f.seek(fob(k, fb))
            f.write(out.tobytes())
            px += w * h
    tot, tmax = reduce_counters(float(px), 1.0 + rank, dist)
    dist.barrier()
    assert tot == float(n_frames * w * h) and tmax == float(world)
    dist.destroy_process_group()


def test_two_rank_frame_shard_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp

    from oracle import binding as ob

    n_frames, world = 5, 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    w, h = 64, 32
    d = ob.make_desc(w, h, dst_depth=10, dst_matrix=ob.MATRIX_BT2020NC, resampler=1)
    want = np.concatenate([oracle.convert_frame(d, oracle.synth_frame(w, h, k)) for k in range(n_frames)])
    got = np.fromfile(os.path.join(str(tmp_path), "out.yuv"), dtype=np.uint16)
    assert np.array_equal(got, want)  # same bytes as one process appending frame after frame


def _run_bench(args, env_extra=None):
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no torchrun around it starts two ranks (rehearsal form: gloo, no device,
    nothing measured): rank r owns frames [r*F, (r+1)*F), the job's time is the slowest rank's, ONE JSON line."""
    r, out = _run_bench(["--gpus", "2", "--rehearse", "--frames", "5", "--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["rehearsal"] is True and out["n_gpus"] == 2 and out["value"] is None
    assert [p["frames"] for p in out["per_rank"]] == [[0, 5], [5, 10]]
    assert out["seconds_max"] == 1.25 and out["pixels_total"] == 2 * 5 * 3840 * 2160 * 3
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1


def test_bench_refuses_fewer_gpus_than_asked():
    """No silent fall-back to one GPU: without N visible devices `--gpus N` exits non-zero and prints no JSON line;
    neither does a WORLD_SIZE that differs from --gpus."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible here")
    r, out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and out is None and "refusing" in r.stderr
    r, out = _run_bench(["--gpus", "1", "--rehearse"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and out is None and "refusing" in r.stderr


# ---- the command line: how unset attributes resolve (hdr2yuv.cpp:61-62, :265-318, :765-766; read_file) -------------

def _cli_dry(args):
    import subprocess

    exe = os.path.join(ROOT, "hdr2yuv_amd", "hdr2yuv")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hdr2yuv_amd", "cli"), "--no-print-directory"], check=True)
    r = subprocess.run([exe] + [str(a) for a in args] + ["--dry_run", "1"], capture_output=True, text=True)
    kv = {}
    for ln in r.stdout.splitlines():
        if ": " in ln and not ln.startswith(("WARNING", "ERROR")):
            k, v = ln.split(": ", 1)
            kv[k] = v
    return r, kv


from cli_lines import TEST_SH  # noqa: E402


def test_cli_resolves_test_sh_lines_like_the_reference(tmp_path):
    """No range flag on any line: in_pic is zeroed (hdr2yuv.cpp:765) so the source range is 0, and the destination copies
    it at parse time (:296-297) -- VIDEO range out, also for the .exr line, where read_exr() forces only the INPUT picture
    to full range afterwards (exr.cpp:183)."""
    for name, line in TEST_SH.items():
        r, kv = _cli_dry(line.format(src=tmp_path / "in", dst=tmp_path / "out").split())
        assert r.returncode == 0, r.stdout
        assert kv["dst_video_full_range_flag"] == "0", name
        assert kv["src_colour_primaries"] == kv["dst_colour_primaries"] == "1"
        assert kv["chroma_resampler_type"] == "1"  # uninitialised in the reference; FIR here
    assert kv["src_full_range_video_flag"] == "0"  # printed before read_exr() would run, as in the reference (:478)


def test_cli_unset_attributes_are_zero_and_dst_copies_src(tmp_path):
    base = ["--src_filename", tmp_path / "a.f32", "--dst_filename", tmp_path / "o.yuv", "--src_pic_width", 64, "--src_pic_height", 32]
    # nothing but geometry and a depth: everything 0, dst depth <- src depth, dst chroma <- the src value AS PARSED (0),
    # while the float input itself is then set to 4:4:4 (hdr2yuv.cpp:351-355)
    r, kv = _cli_dry(base + ["--src_bit_depth", 16])
    assert r.returncode != 0 and "dst_chroma_format_idc must be" in r.stdout  # 0 = mono: resolved as the reference does, not on this path
    for k in ("src_full_range_video_flag", "src_colour_primaries", "src_transfer_characteristics", "src_matrix_coeffs",
              "dst_video_full_range_flag", "dst_colour_primaries", "dst_transfer_characteristics", "dst_matrix_coeffs", "dst_chroma_format_idc"):
        assert kv[k] == "0", k
    assert kv["dst_bit_depth"] == "16" and kv["src_chroma_format_idc"] == "3"
    # dst <- src when only the source is given; an explicit dst wins
    r, kv = _cli_dry(base + ["--src_bit_depth", 16, "--src_video_full_range_flag", 1, "--src_colour_primaries", 9,
                             "--src_transfer_characteristics", 8, "--src_matrix_coeffs", 0, "--dst_transfer_characteristics", 16,
                             "--src_chroma_format_idc", 3, "--dst_chroma_format_idc", 1, "--dst_bit_depth", 10])
    assert (kv["dst_video_full_range_flag"], kv["dst_colour_primaries"], kv["dst_transfer_characteristics"], kv["dst_matrix_coeffs"],
            kv["dst_chroma_format_idc"], kv["dst_bit_depth"]) == ("1", "9", "16", "0", "1", "10")
    # no --src_bit_depth: "src bit_depth(0) outside range [8,32]" -> too many argument errors (hdr2yuv.cpp:531-534)
    r, kv = _cli_dry(base)
    assert r.returncode != 0 and "TOO MANY ARGUMENT ERRORS" in r.stdout and "src bit_depth(0)" in r.stdout
    # integer input without --src_chroma_format_idc: "Only 4:4:4 input supported" (hdr2yuv.cpp:539-543)
    r, kv = _cli_dry(["--src_filename", tmp_path / "a.yuv", "--dst_filename", tmp_path / "o.yuv", "--src_pic_width", 64,
                      "--src_pic_height", 32, "--src_bit_depth", 12])
    assert r.returncode != 0 and "Only 4:4:4 input supported" in r.stdout
    # a codec format names where its decoder lives instead of guessing
    r, kv = _cli_dry(["--src_filename", tmp_path / "a.exr", "--dst_filename", tmp_path / "o.yuv", "--src_pic_width", 64,
                      "--src_pic_height", 32, "--src_bit_depth", 16])
    assert r.returncode != 0 and "exr.cpp" in r.stdout
    # --gpus: the frame blocks of hdr2yuv_amd/shard.py
    r, kv = _cli_dry(["--synthetic", 0, "--dst_filename", tmp_path / "o.yuv", "--src_pic_width", 64, "--src_pic_height", 32, "--src_bit_depth", 32,
                      "--dst_bit_depth", 10, "--dst_chroma_format_idc", 1, "--dst_matrix_coeffs", 9, "--n_frames", 7, "--gpus", 3])
    assert r.returncode == 0 and kv["gpus"] == "3 (devices 0 1 2)" and kv["frames"] == "7" and kv["frame_bytes"] == "6144"


def test_cpu_bench_helper_prints_the_frames_md5s(oracle):
    """oracle/cpu_bench.py is what bench.py's frame-parallel CPU baseline runs, and what its `verified_frames` are compared
    with: one md5 per requested frame index, of the very bytes the oracle gives for that frame (also with a transfer pair)."""
    import hashlib
    import subprocess

    from oracle import binding as ob

    for extra, kw in (([], {}), (["--src-transfer", "16", "--dst-transfer", "1"], dict(src_transfer=16, dst_transfer=1))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py")] + extra + ["64", "32", "1", "12", "9", "3", "0", "11"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        lines = r.stdout.split("\n")
        assert lines[0].split()[0] in ("reference", "port")
        got = {int(ln.split()[1]): ln.split()[2] for ln in lines[1:] if ln.startswith("md5")}
        d = ob.make_desc(64, 32, dst_depth=12, dst_matrix=9, resampler=1, **kw)
        for k in (3, 0, 11):
            assert got[k] == hashlib.md5(oracle.convert_frame(d, synth_frame(64, 32, k)).tobytes()).hexdigest()


def test_bench_counts_gpus_without_touching_them(monkeypatch):
    """The parent of `bench.py --gpus N` must not initialise the GPU (its children are the ranks): devices are counted from the
    visibility variables, else from the KFD topology in sysfs; neither imports torch."""
    import importlib

    sys.modules.pop("bench", None)
    before = "torch" in sys.modules
    bench = importlib.import_module("bench")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,5")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpus()
    assert n is None or n >= 0
    assert ("torch" in sys.modules) == before  # importing bench and counting devices pulled no torch in
