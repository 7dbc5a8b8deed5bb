#!/usr/bin/env python3
"""tools/layoutbench2.py -- placement sweep in ONE pair of arenas (one process, fixed physical pages): the C2 box batch with
plane i of the batch at arena_in + base + i * (plane_bytes + pad_in) and frame f's output at arena_out + base + f * (frame_bytes +
pad_out), for a list of (pad_in, pad_out, base).  Timing experiment only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdr2yuv_amd as h  # noqa: E402


def main():
    F, w, hh = 64, 3840, 2160
    n = w * hh
    res_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    d = h.make_desc(w, hh, dst_depth=12 if not res_kind else 10, dst_matrix=h.MATRIX_BT2020NC, resampler=res_kind)
    nb = h.frame_bytes(d)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    src = [torch.rand(n, device=dev, generator=g) for _ in range(6)]  # six distinct planes, reused round robin
    for t in src:
        t[0], t[1] = 0.0, 1.0
    max_pad = 4 << 20
    arena_in = torch.empty((3 * F * (n * 4 + max_pad) + (8 << 20)) // 4, dtype=torch.float32, device=dev)
    arena_out = torch.empty((F * (nb + max_pad) + (8 << 20)) // 2, dtype=torch.int16, device=dev)
    ctx = h.Context(0)
    cases = []
    stag = [None]
    if len(sys.argv) > 2 and sys.argv[2] == "groups":  # the frame groups' part in it: a bad and a good placement at 1, 2, 4, 8 groups
        for grp in (8, 4, 2, 1):
            for pad in (65536, 262144, 0, 376832):
                cases.append((pad, pad if pad != 376832 else 282624, 0, grp))
    else:
      for pad in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152,
                768, 1280, 4096 + 256, 4096 + 512, 4096 * 3, 4096 * 5, 4096 * 7, 8192 * 3, 65536 + 4096, 65536 * 3, 1048576 + 4096, 1048576 + 65536):
        cases.append((pad, pad, 0, 8))
      for base in (256, 4096, 65536, 1 << 20):
        cases.append((0, 0, base, 8))
        cases.append((4096, 4096, base, 8))
      for pi, po in ((4096, 0), (0, 4096), (4096, 65536), (65536, 4096), (4096, 256), (8192, 4096), (376832, 282624)):
        cases.append((pi, po, 0, 8))
    results = []
    for (pi, po, base, grp) in cases:
        ctx.set_option("groups", grp)
        sin, sout = n * 4 + pi, nb + po
        li, lo = [], []
        for f in range(F):
            fr = []
            for c in range(3):
                o = (base + (f * 3 + c) * sin) // 4
                t = arena_in[o:o + n]
                t.copy_(src[(f * 3 + c) % 6])
                fr.append(t)
            li.append(fr)
            o = (base + f * sout) // 2
            lo.append(arena_out[o:o + nb // 2])
        a = (C.c_void_p * (3 * F))(*[t.data_ptr() for fr in li for t in fr])
        b = (C.c_void_p * F)(*[t.data_ptr() for t in lo])
        torch.cuda.synchronize()
        for st in stag:
            for _ in range(6):
                ctx.convert_batch_enqueue_raw(d, F, a, b)
                ctx.batch_finish()
            tot = 0.0
            for _ in range(15):
                ctx.convert_batch_enqueue_raw(d, F, a, b)
                ctx.batch_finish()
                tot += ctx.last_kernel_ms()[0]
            results.append(((pi, po, base, grp, st), tot / 15))
            print(f"pad_in {pi:8d} pad_out {po:8d} base {base:8d} groups {grp}  {tot / 15:.4f} ms  ({ctx.last_kernel_variant()})", flush=True)
    results.sort(key=lambda r: r[1])
    print("best:", results[:5])
    print("worst:", results[-5:])
    ctx.close()


if __name__ == "__main__":
    main()
