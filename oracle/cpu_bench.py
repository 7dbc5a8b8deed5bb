#!/usr/bin/env python3
"""oracle/cpu_bench.py W H FRAMES RESAMPLER DST_DEPTH DST_MATRIX -- TEST INFRASTRUCTURE ONLY.
One single-threaded process of the CPU reference path (oracle/_ref when it travelled with the repo, else
the C restatement) converting FRAMES synthetic frames; prints "<kind> <seconds>".  bench.py starts one per
host core for its frame-parallel CPU baseline (the reference itself is one process per frame)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdr2yuv_amd.synth import synth_frame  # numpy only
from oracle import binding as ob


def main():
    w, h, n, res, depth, mat = (int(x) for x in sys.argv[1:7])
    try:
        impl, kind = ob.Ref(build=False), "reference"
    except Exception:
        impl, kind = ob.Oracle(), "port"
    d = ob.make_desc(width=w, height=h, dst_depth=depth, dst_matrix=mat, resampler=res)
    planes = synth_frame(w, h, 0)
    t0 = time.perf_counter()
    for _ in range(n):
        impl.convert_frame(d, planes)
    print(kind, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
