import sys, numpy as np
for fn in sys.argv[1:]:
    d = np.loadtxt(fn, dtype=np.int64)
    blk = d[:256]; t0 = blk[:,1].min()
    en = (blk[:,2]-t0)/100.0
    wl = d[800:864]  # wave leave times of blocks 0..3 (entry index 800 + b*16 + w), stored in column 1
    leave = (wl[:,1]-t0)/100.0
    print(fn, "block finish: min %.0f mean %.0f max %.0f" % (en.min(), en.mean(), en.max()), "| waves of blocks 0-3 left the frame loop: min %.0f mean %.0f max %.0f" % (leave.min(), leave.mean(), leave.max()), "| block 0..3 finish", np.round(en[:4]))
