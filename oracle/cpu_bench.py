#!/usr/bin/env python3
"""oracle/cpu_bench.py [--src-transfer T --dst-transfer T] W H RESAMPLER DST_DEPTH DST_MATRIX FRAME [FRAME ...] -- TEST INFRASTRUCTURE ONLY.
One single-threaded process of the CPU reference path (oracle/_ref when it travelled with the repo, else
the C restatement) converting the synthetic frames with the given indices (SURVEY 8c generator, seed
12345 + index); prints "<kind> <seconds>" and one "md5 <index> <md5 of the .yuv frame>" line per frame.
bench.py starts one per host core for its frame-parallel CPU baseline (the reference itself is one process
per frame) and compares the md5s with the GPU's output frames of the same indices."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdr2yuv_amd.synth import synth_frame  # numpy only
from oracle import binding as ob


def main():
    argv = sys.argv[1:]
    tf = {}
    while argv and argv[0] in ("--src-transfer", "--dst-transfer"):
        tf["src_transfer" if argv[0] == "--src-transfer" else "dst_transfer"] = int(argv[1])
        argv = argv[2:]
    w, h, res, depth, mat = (int(x) for x in argv[:5])
    frames = [int(x) for x in argv[5:]]
    try:
        impl, kind = ob.Ref(build=False), "reference"
    except Exception:
        impl, kind = ob.Oracle(), "port"
    d = ob.make_desc(width=w, height=h, dst_depth=depth, dst_matrix=mat, resampler=res, **tf)
    inputs = [synth_frame(w, h, k) for k in frames]
    outs = []
    t0 = time.perf_counter()
    for planes in inputs:
        outs.append(impl.convert_frame(d, planes))
    print(kind, time.perf_counter() - t0, flush=True)
    for k, o in zip(frames, outs):
        print("md5", k, hashlib.md5(o.tobytes()).hexdigest(), flush=True)


if __name__ == "__main__":
    main()
