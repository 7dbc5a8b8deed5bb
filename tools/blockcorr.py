import sys, numpy as np
fs = sys.argv[1:]
devs = []
for f in fs:
    d = np.loadtxt(f, dtype=np.int64); d = d[d[:,2] > 0][:256]
    en = (d[:,2] - d[:,1].min())/100.0
    xm = np.array([en[(d[:,0] % 8) == x].mean() for x in range(8)])
    devs.append(en - xm[d[:,0] % 8])   # deviation from the XCD mean
    print(f, "finish min %.0f mean %.0f max %.0f; within-XCD std %.1f us" % (en.min(), en.mean(), en.max(), devs[-1].std()))
for i in range(len(devs)-1):
    print("corr of within-XCD deviations, launch %d vs %d: %.3f" % (i, i+1, np.corrcoef(devs[i], devs[i+1])[0,1]))
