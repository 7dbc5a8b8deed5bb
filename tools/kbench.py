#!/usr/bin/env python3
"""tools/kbench.py -- kernel-level timing experiments (GPU box). Not part of the product."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hdr2yuv_amd as h
from hdr2yuv_amd.synth import synth_frame


def timeit(ctx, d, frames_in, label, steps=int(os.environ.get("STEPS", "10")), bytes_per_px=15.0):
    frames_in = list(frames_in) * int(os.environ.get("REPEAT", "1"))  # the same inputs again: longer launches without more host work
    F = len(frames_in)
    nb = h.frame_bytes(d)
    outs_t = [torch.empty(nb // 2, dtype=torch.int16, device="cuda") for _ in range(F)]
    ins = (C.c_void_p * (3 * F))(*[t.data_ptr() for fr in frames_in for t in fr])
    outs = (C.c_void_p * F)(*[t.data_ptr() for t in outs_t])
    for _ in range(2):
        ctx.convert_batch_enqueue_raw(d, F, ins, outs)
        ctx.batch_finish()
    kms = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.convert_batch_enqueue_raw(d, F, ins, outs)
        ctx.batch_finish()
        kms += ctx.last_kernel_ms()[0]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    kms /= steps
    px = F * d.width * d.height
    print(f"{label:48s} kernel {kms*1e3/F:8.1f} us/frame  wall {wall*1e3/F:8.1f} us/frame  "
          f"{px/kms/1e6:8.1f} Gpx/s  {px*bytes_per_px/kms/1e6:7.0f} GB/s ({px*bytes_per_px/kms/1e6/80:.1f}% of 8TB/s)", flush=True)


def main():
    F = int(os.environ.get("F", "8"))
    w, hh = 3840, 2160
    ctx = h.Context(0)
    synth = [[torch.from_numpy(p).cuda() for p in synth_frame(w, hh, k)] for k in range(F)]
    which = sys.argv[1:] or ["all"]
    def on(x): return "all" in which or x in which
    if on("c2box"):
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0), synth, "C2 box synthetic")
    if on("c2fir"):
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=1), synth, "C2 FIR synthetic")
    if on("c3"):
        timeit(ctx, h.make_desc(w, hh, dst_depth=16, dst_matrix=11, chroma=3), synth, "C3 444 YDzDx 16b", bytes_per_px=18.0)
    if on("pass"):
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0, src_transfer=16, dst_transfer=16), synth,
               "no PQ (same transfer) box: matrix only")
    if on("const"):
        const = [[torch.full((w * hh,), 0.3 + 0.1 * c, dtype=torch.float32, device="cuda") for c in range(3)] for _ in range(F)]
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0, stats=[(0, 1)] * 3), const,
               "C2 box constant input (LDS broadcast)")
    if on("bars"):
        # letterboxed 2.39:1 in 16:9: 12.8 % black rows at the top and at the bottom; and an all-black frame
        bar = int(hh * 0.128) // 2 * 2
        lb = []
        for k in range(F):
            fr = [p.clone() for p in synth[k]]
            for p in fr:
                p[: bar * w] = 0.0
                p[-bar * w:] = 0.0
                p[bar * w] = 0.0
                p[bar * w + 1] = 1.0
            lb.append(fr)
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0), lb, "C2 box, letterboxed (25.6 % black rows)")
        blk = [[torch.zeros(w * hh, dtype=torch.float32, device="cuda") for _ in range(3)] for _ in range(F)]
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0, stats=[(0, 1)] * 3), blk, "C2 box, all-black frames")
    if on("smooth"):
        # smooth ramp: neighbouring pixels share table segments, like natural images
        ramp = (torch.arange(w * hh, device="cuda", dtype=torch.float32) % w) / w
        sm = [[(ramp * (0.5 + 0.1 * c)).contiguous() for c in range(3)] for _ in range(F)]
        for fr in sm:
            for p in fr:
                p[0] = 0.0
                p[1] = 1.0
        timeit(ctx, h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0), sm, "C2 box smooth ramp input")
    if on("u16"):
        # 16-bit integer input, equal transfers (the .tiff/.yuv flows): no PQ, matrix + subsample only
        rng = np.random.default_rng(3)
        u = [[torch.from_numpy(rng.integers(0, 65536, w * hh, dtype=np.uint16).view(np.int16)).cuda() for _ in range(3)] for _ in range(F)]
        timeit(ctx, h.make_desc(w, hh, sample=h.SAMPLE_U16, src_depth=16, dst_depth=10, dst_matrix=9, resampler=0, src_transfer=16, dst_transfer=16), u,
               "4K u16 RGB -> 10-bit 2020nc 4:2:0 box, equal transfers", bytes_per_px=9.0)
    if on("inv"):
        # matrix_inverse, 4K 12-bit 4:4:4 BT.709 -> 16-bit G,B,R planes; 12 B/px
        rng = np.random.default_rng(4)
        n = w * hh
        din = [torch.from_numpy(rng.integers(0, 4096, n, dtype=np.uint16).view(np.int16)).cuda() for _ in range(3)]
        dout = [torch.empty(n, dtype=torch.int16, device="cuda") for _ in range(3)]
        for mat, name in ((1, "BT.709"), (11, "YDzDx")):
            ms = 0.0
            for _ in range(10):
                ctx.matrix_inverse(w, hh, 12, 0, mat, 16, din, dout)
                ms += ctx.last_kernel_ms()[0]
            ms /= 10
            print(f"matrix_inverse 4K 12-bit {name:7s} -> 16-bit GBR   kernel {ms*1e3:8.1f} us/frame  {n/ms/1e6:8.1f} Gpx/s  {n*12/ms/1e6:7.0f} GB/s ({n*12/ms/1e6/80:.1f}% of 8TB/s)", flush=True)
    if on("c1"):
        w1, h1 = 1920, 1080
        s1 = [[torch.from_numpy(p).cuda() for p in synth_frame(w1, h1, k)] for k in range(F)]
        timeit(ctx, h.make_desc(w1, h1, dst_depth=12, dst_matrix=9, resampler=0), s1, "1080p 12-bit 2020nc box (cache-resident when F small)")
    if on("c4"):
        w8, h8 = 7680, 4320
        f16 = [[torch.from_numpy(p.view(np.int16)).cuda() for p in synth_frame(w8, h8, k, f16=True)] for k in range(max(2, F // 4))]
        timeit(ctx, h.make_desc(w8, h8, sample=3, dst_depth=10, dst_matrix=9, resampler=0), f16, "C4 8K f16 box", bytes_per_px=9.0)
    ctx.close()


if __name__ == "__main__":
    main()
