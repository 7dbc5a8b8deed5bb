#!/usr/bin/env python3
"""tools/copyoverlap.py -- do an upload and a download on two streams overlap on this box?  (the
pipeline's H2D of frame k+1 and D2H of frame k-1.)  Pinned host buffers of a C2 frame's sizes (99.5 MB up, 24.9 MB
down), 20 rounds each way: alone, together on two streams, and together through hipMemcpyAsync by ctypes (no torch)."""
import time

import torch

up_n, dn_n = 99_532_800, 24_883_200
h_up = torch.empty(up_n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(dn_n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(up_n, dtype=torch.uint8, device="cuda")
d_dn = torch.empty(dn_n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
R = 20


def run(up, dn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        if up:
            with torch.cuda.stream(s1):
                d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2):
                h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R * 1e3


for _ in range(2):
    run(True, True)
a, b, c = run(True, False), run(False, True), run(True, True)
print(f"upload alone {a:.3f} ms ({up_n / a / 1e6:.1f} GB/s), download alone {b:.3f} ms ({dn_n / b / 1e6:.1f} GB/s), both on two streams {c:.3f} ms "
      f"(sum {a + b:.3f}, max {max(a, b):.3f}): {'overlap' if c < 0.9 * (a + b) else 'NO overlap'}")
