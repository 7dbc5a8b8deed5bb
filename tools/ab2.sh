#!/bin/bash
# tools/ab2.sh OUTDIR -- in one gpurun call: the in-tree library against every build/ab/*.so on the box and FIR paths of C2 with
# the three --content pictures (bench.py, 20 steps), two rounds alternating.
out=$1; mkdir -p "$out"
libs="intree $(ls build/ab/*.so 2>/dev/null)"
for r in 1 2; do
  for c in uniform bars squared; do
    for res in box fir; do
      for l in $libs; do
        n=$(basename "$l" .so)
        if [ "$l" = intree ]; then extra=""; else extra="--lib $l"; fi
        python bench.py --no-extra --no-cpu-baseline --content $c --resampler $res --steps 20 --warmup 10 $extra > "$out/${res}_${c}_${n}_$r.json" 2> "$out/${res}_${c}_${n}_$r.err" || echo "FAILED $res $c $n $r"
      done
    done
  done
done
python - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.load(open(f)); r = j["roofline"]
        print(f"{os.path.basename(f):36s} {j['value']:10.1f} {r['kernel_ms_per_step']:.4f} {j['verified']} {r['variant']}")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
