#!/bin/bash
# tools/ab_libs.sh OUTDIR "LIB1 LIB2 ..." [common bench args] -- several builds of the library on one box: 6 rounds, the order rotated
# every round, 40 steps each; prints every run's kernel ms and the medians.  "-" stands for the in-tree library.
out=$1; libs=($2); shift 2
mkdir -p "$out"
n=${#libs[@]}
for r in 0 1 2 3 4 5; do
  for k in $(seq 0 $((n-1))); do
    i=$(( (k + r) % n )); lib=${libs[$i]}
    if [ "$lib" = "-" ]; then extra=""; else extra="--lib $lib"; fi
    python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 "$@" $extra > "$out/r${r}_lib${i}.json" 2>/dev/null || echo "FAILED $r $lib"
  done
done
python - "$out" "${libs[@]}" <<'PY'
import glob, json, os, sys, statistics
out, libs = sys.argv[1], sys.argv[2:]
for i, lib in enumerate(libs):
    v = []
    for f in sorted(glob.glob(os.path.join(out, f"r*_lib{i}.json"))):
        j = json.load(open(f)); v.append(j["roofline"]["kernel_ms_per_step"])
        if not j["verified"]: print("NOT VERIFIED", f)
    print(f"{lib:40s}", " ".join(f"{x:.4f}" for x in v), "median", round(statistics.median(v), 4), "min", min(v))
PY
