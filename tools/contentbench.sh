# tools/contentbench.sh [OUTDIR] -- bench.py --content uniform | bars | squared on the box and FIR paths of C2
set -e
o=${1:-gpurun_out/content}; mkdir -p $o
for c in uniform bars squared; do
  python bench.py --no-cpu-baseline --no-extra --content $c --steps 20 --warmup 10 > $o/box_$c.json 2> $o/box_$c.err
  python bench.py --no-cpu-baseline --no-extra --content $c --resampler fir --steps 20 --warmup 10 > $o/fir_$c.json 2> $o/fir_$c.err
done
python bench.py --no-cpu-baseline --no-extra --option t1=0 --steps 20 --warmup 10 > $o/box_uniform_t1off.json 2> $o/box_uniform_t1off.err
python bench.py --no-cpu-baseline --no-extra --option t1=0 --resampler fir --steps 20 --warmup 10 > $o/fir_uniform_t1off.json 2> $o/fir_uniform_t1off.err
python - $o <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j = json.load(open(f)); r = j["roofline"]
    print(f"{os.path.basename(f):28s} {j['value']:10.1f} kernel {r['kernel_ms_per_step']:.4f} ms frac {r['frac']:.4f} {r['variant']} redone {j.get('frames_redone')}")
PY
