import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import hdr2yuv_amd as h
from oracle import binding as ob
import test_gpu_parity as T
ctx = h.Context(0)
orc = ob.Oracle()
for depth, mat, chroma, res in [(12, 9, 1, 0), (16, h.MATRIX_YDZDX, h.CHROMA_444, 0)]:
    rng = np.random.default_rng(77 + depth)
    w, hh = 264, 80
    planes = T._picture_like(rng, w, hh)
    for stats in ([(0, 1)] * 3, None):
        d = h.make_desc(w, hh, dst_depth=depth, dst_matrix=mat, chroma=chroma, resampler=res, stats=stats)
        got = ctx.convert_frame(d, planes)
        want = orc.convert_frame(T._to_oracle_desc(d), planes)
        bad = np.nonzero(got != want)[0]
        print(depth, mat, chroma, res, "stats", stats is not None, "bad", len(bad))
        n = w * hh
        for i in bad[:12]:
            if i < n:
                y, x = divmod(int(i), w)
                print("  Y at", y, x, "got", got[i], "want", want[i], "in", [float(p[i]) for p in planes])
            else:
                j = int(i) - n
                cw = w // 2 if chroma == 1 else w
                pl, j2 = divmod(j, (cw * (hh // 2 if chroma == 1 else hh)))
                y, x = divmod(j2, cw)
                print("  C%d at" % pl, y, x, "got", got[i], "want", want[i])
ctx.close()
