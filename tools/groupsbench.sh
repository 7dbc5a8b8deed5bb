# tools/groupsbench.sh -- the full bench line with the frame groups chosen by frame size (default), eight and one, three rounds alternating
o=gpurun_out/r2q; mkdir -p $o
for r in 1 2 3; do
  python bench.py --no-cpu-baseline > $o/auto_$r.json 2>/dev/null || echo FAILED auto
  python bench.py --no-cpu-baseline --option groups=8 > $o/g8_$r.json 2>/dev/null || echo FAILED g8
  python bench.py --no-cpu-baseline --option groups=1 > $o/g1_$r.json 2>/dev/null || echo FAILED g1
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r2q/*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), j["roofline"]["frac"], " ".join(f"{k}={v['frac']}{v['kernel_ms_min_max']}" for k, v in j["others"].items()), j["verified"], all(v["verified"] for v in j["others"].values()))
PY
