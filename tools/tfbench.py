#!/usr/bin/env python3
"""tools/tfbench.py -- time the transfer pairs other than LINEAR -> PQ: they run through the generic
kernel's careful tier (double-double pow per sample).  us per 4K frame, 8 frames per launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hdr2yuv_amd as h  # noqa: E402
from hdr2yuv_amd.synth import synth_frame  # noqa: E402

w, hh, n = 3840, 2160, 8
frames = [[torch.from_numpy(p).cuda() for p in synth_frame(w, hh, k)] for k in range(n)]
ctx = h.Context(0)
for (src, dst, name) in ((8, 16, "LINEAR->PQ (first tier)"), (16, 8, "PQ->LINEAR"), (1, 16, "BT.709->PQ"), (8, 1, "LINEAR->BT.709"), (18, 16, "RHO_GAMMA->PQ"),
                         (8, 18, "LINEAR->RHO_GAMMA"), (16, 1, "PQ->BT.709")):
    d = h.make_desc(w, hh, dst_depth=12, src_transfer=src, dst_transfer=dst, dst_matrix=9, resampler=0, stats=[(0, 1)] * 3)
    outs = [torch.empty(h.frame_bytes(d) // 2, dtype=torch.int16, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.convert_batch(d, frames, outs)
    ms, launches = ctx.last_kernel_ms()
    print(f"{name:28s} {ms / n * 1e3:9.1f} us/frame  {w * hh * n / ms / 1e6:8.2f} Gpixel/s  {ctx.last_kernel_variant()}")
ctx.close()
