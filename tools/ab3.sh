#!/bin/bash
# tools/ab3.sh OUTDIR LIB_B [bench args] -- careful A/B of the in-tree library (A) against LIB_B: ABBA order, 6 rounds, 40 steps each;
# prints every run's kernel ms and the medians.  (Runs of one library on one box differ by several per cent from minute to minute.)
out=$1; libb=$2; shift 2
mkdir -p "$out"
i=0
for r in 1 2 3 4 5 6; do
  if [ $((r % 2)) = 1 ]; then order="A B"; else order="B A"; fi
  for w in $order; do
    i=$((i+1))
    if [ $w = A ]; then extra=""; else extra="--lib $libb"; fi
    python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 "$@" $extra > "$out/$(printf %02d $i)_$w.json" 2>/dev/null || echo "FAILED $i $w"
  done
done
python - "$out" <<'PY'
import glob, json, os, sys, statistics
v = {"A": [], "B": []}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j = json.load(open(f)); w = os.path.basename(f)[3]
    v[w].append(j["roofline"]["kernel_ms_per_step"])
for w in "AB":
    print(w, " ".join(f"{x:.4f}" for x in v[w]), "median", round(statistics.median(v[w]), 4), "min", min(v[w]))
PY
