#!/usr/bin/env python3
"""tools/streambench.py -- end-to-end frames/s from HOST memory, GPU box.
One 4K C2 frame = 99.5 MB up, 24.9 MB down: PCIe-bound.  Compares the synchronous entry
(h2y_convert_frame: upload, convert, download one after the other, pageable numpy buffers) with the
pinned ring of h2y_stream_* (the three overlapped)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hdr2yuv_amd as h
from hdr2yuv_amd.synth import synth_frame


def main():
    n = int(os.environ.get("N", "200"))  # long enough for the start-up (three slots filled by host copies) not to weigh
    depth = int(os.environ.get("DEPTH", "3"))
    w, hh = 3840, 2160
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=9, resampler=0)
    planes = synth_frame(w, hh, 0)
    ctx = h.Context(0)
    ctx.convert_frame(d, planes)
    t0 = time.perf_counter()
    for _ in range(n // 4):
        ctx.convert_frame(d, planes)
    dt = (time.perf_counter() - t0) / (n // 4)
    print(f"h2y_convert_frame (pageable host buffers, serial):   {dt*1e3:7.2f} ms/frame  {1/dt:7.1f} frames/s", flush=True)

    ctx.stream_open(d, depth)
    for fill in (False, True):
        # fill=True also pays for writing the input into the pinned slot (a memcpy standing in for the file read)
        inflight = 0
        done = 0
        t0 = time.perf_counter()
        for _ in range(n):
            dst = ctx.stream_input()
            if fill or done + inflight < depth:
                for c in range(3):
                    dst[c][:] = planes[c]
            ctx.stream_submit()
            inflight += 1
            if inflight == depth - 1:
                ctx.stream_output()
                inflight -= 1
                done += 1
        while inflight:
            ctx.stream_output()
            inflight -= 1
            done += 1
        dt = (time.perf_counter() - t0) / n
        gb = (3 * w * hh * 4 + h.frame_bytes(d)) / 1e9
        print(f"h2y_stream_* depth {depth}{' + host copy into the slot' if fill else '':28s}: {dt*1e3:7.2f} ms/frame  {1/dt:7.1f} frames/s  ({gb/dt:5.1f} GB/s over PCIe)", flush=True)
    ctx.stream_close()
    ctx.close()


if __name__ == "__main__":
    main()
