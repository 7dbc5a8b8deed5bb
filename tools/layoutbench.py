#!/usr/bin/env python3
"""tools/layoutbench.py -- does the placement of the planes in device memory matter?  One process, one context: the C2 box
batch (64 x 4K fp32 -> 12-bit 4:2:0) with its 192 input planes and 64 output frames carved out of ONE allocation each at
different paddings between consecutive planes / frames, against separately allocated tensors (what bench.py does), each
layout timed several times in alternation.  Timing experiment only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdr2yuv_amd as h  # noqa: E402


def main():
    F, w, hh = 64, 3840, 2160
    n = w * hh
    d = h.make_desc(w, hh, dst_depth=12, dst_matrix=h.MATRIX_BT2020NC, resampler=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    nb = h.frame_bytes(d)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    layouts = {}
    # separately allocated tensors
    ins = [[torch.rand(n, device=dev, generator=g) for _ in range(3)] for _ in range(F)]
    outs = [torch.empty(nb // 2, dtype=torch.int16, device=dev) for _ in range(F)]
    for fr in ins:
        fr[0][0], fr[0][1] = 0.0, 1.0
        fr[1][0], fr[1][1] = 0.0, 1.0
        fr[2][0], fr[2][1] = 0.0, 1.0
    layouts["separate tensors"] = (ins, outs, None)
    for pad in (0, 256, 4096, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 8192 * 3 + 512):
        pin = (n * 4 + pad + 255) // 256 * 256
        pout = (nb + pad + 255) // 256 * 256
        bi = torch.empty(pin * 3 * F // 4 + 64, dtype=torch.float32, device=dev)
        bo = torch.empty(pout * F // 2 + 64, dtype=torch.int16, device=dev)
        li, lo = [], []
        for f in range(F):
            fr = []
            for c in range(3):
                o = (f * 3 + c) * pin // 4
                t = bi[o:o + n]
                t.copy_(ins[f][c])
                fr.append(t)
            li.append(fr)
            lo.append(bo[f * pout // 2: f * pout // 2 + nb // 2])
        layouts[f"one allocation, pad {pad} B"] = (li, lo, (bi, bo))
    ctx = h.Context(0)
    arrs = {}
    for name, (li, lo, _) in layouts.items():
        a = (C.c_void_p * (3 * F))(*[t.data_ptr() for fr in li for t in fr])
        b = (C.c_void_p * F)(*[t.data_ptr() for t in lo])
        arrs[name] = (a, b)
    torch.cuda.synchronize()
    res = {k: [] for k in layouts}
    for grp in (8, 4, 2, 1, 8):  # frame groups: how many frames the card works on at once
        ctx.set_option("groups", grp)
        for name, (a, b) in arrs.items():
            for _ in range(6):
                ctx.convert_batch_enqueue_raw(d, F, a, b)
                ctx.batch_finish()
            tot = 0.0
            for _ in range(20):
                ctx.convert_batch_enqueue_raw(d, F, a, b)
                ctx.batch_finish()
                tot += ctx.last_kernel_ms()[0]
            res[name].append(tot / 20)
    print(f"{'kernel ms per launch at frame groups':44s}      8      4      2      1      8")
    for name, v in res.items():
        print(f"{name:44s} " + " ".join(f"{x:.4f}" for x in v) + f"   ({ctx.last_kernel_name()})")
    ctx.close()


if __name__ == "__main__":
    main()
