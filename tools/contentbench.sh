set -e
o=gpurun_out/r2m; mkdir -p $o
for c in uniform bars squared; do
  python bench.py --no-cpu-baseline --no-extra --content $c --steps 20 --warmup 10 > $o/box_$c.json 2> $o/box_$c.err
  python bench.py --no-cpu-baseline --no-extra --content $c --resampler fir --steps 20 --warmup 10 > $o/fir_$c.json 2> $o/fir_$c.err
done
python bench.py --no-cpu-baseline --no-extra --option t1=0 --steps 20 --warmup 10 > $o/box_uniform_t1off.json 2> $o/box_uniform_t1off.err
python bench.py --no-cpu-baseline --no-extra --option t1=0 --resampler fir --steps 20 --warmup 10 > $o/fir_uniform_t1off.json 2> $o/fir_uniform_t1off.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r2m/*.json")):
    j = json.load(open(f)); r = j["roofline"]
    print(f"{os.path.basename(f):28s} {j['value']:10.1f} kernel {r['kernel_ms_per_step']:.4f} ms frac {r['frac']:.4f} {r['variant']} redone {j.get('frames_redone')}")
PY
