#!/bin/bash
# tools/ab.sh OUTDIR ARGS... -- A/B timing of library builds in ONE gpurun call (boxes differ by several per cent):
# runs `bench.py ARGS --no-extra --no-cpu-baseline` with the in-tree library and with every build/ab/*.so, three rounds
# alternating, and prints value / kernel ms per step of each run.
out=$1; shift
mkdir -p "$out"
libs="intree $(ls build/ab/*.so 2>/dev/null)"
for r in 1 2 3; do
  for l in $libs; do
    n=$(basename "$l" .so)
    if [ "$l" = intree ]; then extra=""; else extra="--lib $l"; fi
    python bench.py "$@" --no-extra --no-cpu-baseline $extra > "$out/${n}_$r.json" 2> "$out/${n}_$r.err" || echo "FAILED $n $r"
  done
done
python - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.load(open(f))
        print(os.path.basename(f), j["value"], j["roofline"]["kernel_ms_per_step"], j["verified"], j["roofline"]["variant"])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
