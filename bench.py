#!/usr/bin/env python3
"""bench.py -- throughput of the hdr2yuv convert hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (pic_stats -> matrix_convert/PQ -> convert
-> write_yuv arithmetic) over one batch of --frames synthetic frames per GPU,
inputs already resident in HBM, outputs left in HBM in .yuv layout.  Workload =
BASELINE.json configs[1]: 3840x2160 fp32 RGB -> PQ -> 12-bit BT.2020nc YCbCr
4:2:0 (video range), chroma by the 2x2 box (the north_star's fused kernel);
the FIR resampler (make.sh's example) is measured too and reported beside it.

Frames shard by frame index: rank r owns frames [r*F, (r+1)*F) -- no data-path
collective; RCCL carries one all-reduce of the counters (weak scaling).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak (spec); ~5600 measured with a plain float4 copy

WORKLOADS = {
    # name: (desc kwargs, algorithmic bytes per pixel: read input once + write output once)
    "C2": (dict(width=3840, height=2160, dst_depth=12, dst_matrix=9), 15.0,
           "3840x2160 fp32 RGB -> PQ -> 12-bit BT.2020nc YCbCr 4:2:0"),
    "C3": (dict(width=3840, height=2160, dst_depth=16, dst_matrix=11, chroma=3), 18.0,
           "3840x2160 fp32 XYZ -> PQ -> 16-bit YDzDx 4:4:4"),
    "C4": (dict(width=7680, height=4320, sample=3, dst_depth=10, dst_matrix=9), 9.0,
           "7680x4320 fp16 RGB -> PQ -> 10-bit BT.2020nc YCbCr 4:2:0"),
    "C1": (dict(width=1920, height=1080, dst_depth=10, dst_matrix=1), 15.0,
           "1920x1080 fp32 RGB -> PQ -> 10-bit BT.709 YCbCr 4:2:0"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)  # the card clocks up during the first ~10 launches
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step (one launch covers them all; 64 x 124 MB = 8 GB of the 288)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--resampler", default="box", choices=["box", "fir"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (other resampler) measurement")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames for the CPU baseline sample (0 = auto ~15 s)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def run_steps(ctx, d, n_frames, ins, outs, steps, warmup, barrier):
    """W untimed + K timed steps. Returns (seconds for K steps, mean kernel ms per step, redone frames)."""
    import torch

    for _ in range(warmup):
        ctx.convert_batch_enqueue_raw(d, n_frames, ins, outs)
        ctx.batch_finish()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = 0.0
    redone = 0
    for _ in range(steps):
        ctx.convert_batch_enqueue_raw(d, n_frames, ins, outs)
        redone += ctx.batch_finish()
        ms, _n = ctx.last_kernel_ms()
        kms += ms
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, kms / max(steps, 1), redone


def cpu_baseline(desc_kw, resampler, n_frames_hint):
    """The reference CPU path timed on this box's host cores, single thread (the
    reference is single-threaded): oracle/_ref (the reference's own object code)
    when that prebuilt library travelled with the repo, else our C restatement."""
    import numpy as np  # noqa: F401

    from hdr2yuv_amd.synth import synth_frame
    from oracle import binding as ob

    kind = "port"
    impl = None
    try:
        impl = ob.Ref(build=False)
        kind = "reference"
    except Exception:
        impl = ob.Oracle()
    kw = dict(desc_kw)
    kw["resampler"] = 1 if resampler == "fir" else 0
    od = ob.make_desc(**kw)
    w, h = kw["width"], kw["height"]
    planes = synth_frame(w, h, 0, f16=kw.get("sample") == 3)
    t0 = time.perf_counter()
    impl.convert_frame(od, planes)
    one = time.perf_counter() - t0
    n = n_frames_hint or max(1, min(8, int(15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for k in range(n):
        impl.convert_frame(od, planes)
    dt = time.perf_counter() - t0
    return {
        "value": round(n * w * h / dt / 1e6, 3),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{n} frame(s) of the same {w}x{h} workload ({resampler}), {dt:.1f} s, single thread, "
                  f"pic_stats+matrix_convert+convert+write_yuv arithmetic",
    }


def cpu_baseline_parallel(desc_kw, resampler, frames_each=2):
    """Frame-parallel run of the same CPU path: one single-threaded process per host core, each converting
    `frames_each` frames (SURVEY 8d asks for both figures; the reference itself is one process per frame)."""
    import subprocess

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("H2Y_CPU_WORKERS", "16"))))  # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says
    kw = dict(desc_kw)
    if kw.get("sample") == 3 or kw.get("chroma", 1) != 1:
        return None  # the helper script covers the fp32 4:2:0 workloads
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), str(kw["width"]), str(kw["height"]), str(frames_each),
           "1" if resampler == "fir" else "0", str(kw["dst_depth"]), str(kw["dst_matrix"])]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
    kinds = set()
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode != 0:
            return None
        kinds.add(out.split()[0])
    dt = time.perf_counter() - t0  # includes the start-up of the processes: a lower bound of the rate
    return {"value": round(cores * frames_each * kw["width"] * kw["height"] / dt / 1e6, 2), "unit": "Mpixels/s", "cores": cores,
            "kind": "reference" if kinds == {"reference"} else "port",
            "sample": f"{cores} processes x {frames_each} frame(s), {dt:.1f} s wall including process start-up"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import hdr2yuv_amd as h
    from hdr2yuv_amd.shard import frames_for_rank
    from hdr2yuv_amd.synth import synth_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    n_gpus = world
    if rank == 0 and args.gpus != world:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    def barrier():
        if world > 1:
            dist.barrier()

    desc_kw, bytes_per_px, wl_name = WORKLOADS[args.workload]
    is420 = desc_kw.get("chroma", 1) == 1
    F = args.frames
    w, hh = desc_kw["width"], desc_kw["height"]
    f16 = desc_kw.get("sample") == 3
    dev = torch.device("cuda", local_rank)

    # ---- synthetic input, resident in HBM before any timed region --------
    frames_in = []
    for k in frames_for_rank(world * F, rank, world):  # this rank's contiguous block of the global frame sequence
        planes = synth_frame(w, hh, k, f16=f16)
        frames_in.append([torch.from_numpy(p.view(np.int16) if f16 else p).to(dev) for p in planes])
    ctx = h.Context(local_rank)

    def make_io(resampler):
        d = h.make_desc(**dict(desc_kw, resampler=1 if resampler == "fir" else 0))
        nb = h.frame_bytes(d)
        outs_t = [torch.empty(nb // 2, dtype=torch.int16, device=dev) for _ in range(F)]
        ins = (C.c_void_p * (3 * F))(*[t.data_ptr() for fr in frames_in for t in fr])
        outs = (C.c_void_p * F)(*[t.data_ptr() for t in outs_t])
        return d, ins, outs, outs_t

    # a plain device-to-device copy of 1 GiB on this very box (30 times), as the practical HBM ceiling beside the
    # 8 TB/s spec; measured first, while the inputs are fresh in HBM and before the W warm-up steps
    copy_gbs = None
    try:
        src_t = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst_t = torch.empty_like(src_t)
        for _ in range(2):
            dst_t.copy_(src_t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            dst_t.copy_(src_t)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 30 * 2 * src_t.numel() * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9
        del src_t, dst_t
    except Exception:
        copy_gbs = None
    results = {}
    order = [args.resampler] + ([] if (args.no_extra or not is420) else [r for r in ("box", "fir") if r != args.resampler])
    for res in order:
        d, ins, outs, outs_t = make_io(res)
        secs, kernel_ms, redone = run_steps(ctx, d, F, ins, outs, args.steps, args.warmup, barrier)
        t = torch.tensor([secs], dtype=torch.float64, device=dev)
        px = torch.tensor([float(F) * w * hh * args.steps], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)   # max over ranks
            dist.all_reduce(px, op=dist.ReduceOp.SUM)  # pixels all ranks processed (RCCL over xGMI)
        results[res] = dict(secs=float(t.item()), pixels=float(px.item()), kernel_ms=kernel_ms, redone=redone, kernel=ctx.last_kernel_name(),
                            checksum=int(outs_t[0][:4096].to(torch.int64).sum().item()))
        del outs_t

    main_r = results[args.resampler]
    value = main_r["pixels"] / main_r["secs"] / 1e6
    ms_per_step = main_r["secs"] / args.steps * 1e3

    out = {
        "metric": "Mpixels/s, 4K RGB->YUV420 PQ convert path (in-memory, HBM-resident)",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32+f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {wl_name}, chroma {args.resampler}" if is420 else f"{args.workload}: {wl_name}",
            "frames_per_gpu_per_step": F,
            "frame_shard": "frame index, contiguous block per rank",
            "resampler": args.resampler if is420 else "none",
        },
        "frames_per_s": round(value * 1e6 / (w * hh), 1),
    }
    # ---- roofline of the dominant kernel (k_fused), HIP-event timed on its stream
    alg_bytes = bytes_per_px * w * hh * F  # per launch: one launch covers the F frames of a step (box / 4:4:4)
    kernel_s = main_r["kernel_ms"] / 1e3
    launches = 1 if (args.resampler == "box" or not is420) else (F + 31) // 32
    ach = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"{args.workload}_{args.resampler}_F{F}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out["roofline"] = {
        "bound": "hbm",
        "kernel": main_r["kernel"],
        "achieved": round(ach, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": alg_bytes / launches,
        "launches_per_step": launches,
        "kernel_ms_per_step": round(main_r["kernel_ms"], 4),
        "device_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1),
        "frac_of_device_copy": None if not copy_gbs else round(ach / copy_gbs, 4),
    }
    for res, r in results.items():
        if res != args.resampler:
            out[f"{res}_value"] = round(r["pixels"] / r["secs"] / 1e6, 1)
            out[f"{res}_ms_per_step"] = round(r["secs"] / args.steps * 1e3, 4)
    out["frames_redone"] = main_r["redone"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(desc_kw, args.resampler, args.cpu_frames)
            par = cpu_baseline_parallel(desc_kw, args.resampler)
            if par:
                out["cpu_baseline_all_cores"] = par
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
