#!/bin/bash
# tools/ab_others.sh OUTDIR LIB_B -- the in-tree library (A) against another build (B) on the loop-form kernels beside the headline:
# C3 (k_fused2), C4 box (k_fused_lut16), a transfer pair (k_fused2<..,TFN>); 4 rounds, ABBA
out=$1; libb=$2; mkdir -p "$out"
for r in 1 2 3 4; do
  if [ $((r % 2)) = 1 ]; then order="A B"; else order="B A"; fi
  for w in $order; do
    if [ $w = A ]; then extra=""; else extra="--lib $libb"; fi
    python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 10 --workload C3 $extra > "$out/c3_${r}_$w.json" 2>/dev/null || echo FAILED c3
    python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 10 --workload C4 --frames 16 $extra > "$out/c4_${r}_$w.json" 2>/dev/null || echo FAILED c4
    python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 10 --workload tf_bt709_to_pq $extra > "$out/tf_${r}_$w.json" 2>/dev/null || echo FAILED tf
  done
done
python - "$out" <<'PY'
import glob, json, os, sys, statistics
for k in ("c3", "c4", "tf"):
    for w in "AB":
        v = []
        for f in sorted(glob.glob(os.path.join(sys.argv[1], f"{k}_*_{w}.json"))):
            j = json.load(open(f)); v.append(j["roofline"]["kernel_ms_per_step"])
            if not j["verified"]: print("NOT VERIFIED", f)
        if v: print(k, w, " ".join(f"{x:.4f}" for x in v), "median", round(statistics.median(v), 4))
PY
