#!/bin/bash
# tools/ab_opt.sh OUTDIR "ARGS_A" "ARGS_B" [common bench args] -- A/B of two bench.py argument sets on one box: ABBA order, 6 rounds, 40 steps
# each; prints every run's kernel ms and the medians.  (Runs of one library on one box differ by several per cent from process to process.)
out=$1; argsa=$2; argsb=$3; shift 3
mkdir -p "$out"
i=0
for r in 1 2 3 4 5 6; do
  if [ $((r % 2)) = 1 ]; then order="A B"; else order="B A"; fi
  for w in $order; do
    i=$((i+1))
    if [ $w = A ]; then extra="$argsa"; else extra="$argsb"; fi
    python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 "$@" $extra > "$out/$(printf %02d $i)_$w.json" 2>/dev/null || echo "FAILED $i $w"
  done
done
python - "$out" <<'PY'
import glob, json, os, sys, statistics
v = {"A": [], "B": []}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j = json.load(open(f)); w = os.path.basename(f)[3]
    v[w].append(j["roofline"]["kernel_ms_per_step"])
    if not j["verified"]: print("NOT VERIFIED", f)
for w in "AB":
    print(w, " ".join(f"{x:.4f}" for x in v[w]), "median", round(statistics.median(v[w]), 4), "min", min(v[w]))
PY
