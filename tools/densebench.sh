#!/bin/bash
# tools/densebench.sh OUTDIR -- where does the first tier stop paying?  Pictures with a growing share of samples below its table
# (u^2, u^3, u^4, u^6 of the uniform picture), box and FIR: the first-tier kernels held on ("t1=always") against the binary64
# tier's ("t1=0"); the share of pixels passed on is in the variant string (flagged=)
o=${1:-gpurun_out/dense}; mkdir -p $o
for c in squared pow3 pow4 pow6; do
  for res in box fir; do
    for t in always 0; do
      python bench.py --no-cpu-baseline --no-extra --content $c --resampler $res --option t1=$t --steps 12 --warmup 6 > $o/${res}_${c}_t1$t.json 2>> $o/err.txt
    done
  done
done
python - $o <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j = json.load(open(f)); r = j["roofline"]
    print(f"{os.path.basename(f):28s} kernel {r['kernel_ms_per_step']:.4f} ms frac {r['frac']:.4f} {r['variant']}")
PY
