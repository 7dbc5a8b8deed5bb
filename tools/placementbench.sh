o=gpurun_out/r2p; mkdir -p $o
for r in 1 2 3; do
 for pl in arena separate; do
  for g in 8 2 1; do
   for res in box fir; do
     [ $res = fir ] && [ $g != 8 ] && continue
     python bench.py --no-extra --no-cpu-baseline --placement $pl --option groups=$g --resampler $res > $o/${res}_${pl}_g${g}_$r.json 2>/dev/null || echo FAILED $pl $g $res
   done
  done
 done
done
python - <<'PY'
import glob, json, os, collections
v = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r2p/*.json")):
    j = json.load(open(f)); k = os.path.basename(f).rsplit("_", 1)[0]
    v[k].append((j["roofline"]["kernel_ms_per_step"], j["verified"]))
for k, x in sorted(v.items()):
    print(f"{k:24s}", " ".join(f"{a:.4f}" for a, _ in x), all(b for _, b in x))
PY
