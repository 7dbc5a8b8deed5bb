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
