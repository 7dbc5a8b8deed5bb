#!/usr/bin/env python3
"""tools/blocktimes.py -- summarise the per-block start/finish times of a k_fused_t1 launch (library built with -DH2Y_BLOCK_TIMES)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
d = d[d[:, 2] > 0]
t0 = d[:, 1].min()
st, en = (d[:, 1] - t0) / 100.0, (d[:, 2] - t0) / 100.0  # 100 MHz -> us
print(f"blocks {len(d)}  start spread {st.max():.1f} us  finish: min {en.min():.1f} median {np.median(en):.1f} max {en.max():.1f} us")
for x in range(8):
    m = (d[:, 0] % 8) == x
    print(f"  XCD {x}: finish min {en[m].min():8.1f} mean {en[m].mean():8.1f} max {en[m].max():8.1f}")
