# tools/contentbench2.sh OUTDIR -- squared / uniform content on the FIR path with the waves in step (firsync=2) and out of step (0), and the box path
o=${1:-gpurun_out/content2}; mkdir -p $o
for c in squared uniform; do
  for fs in 0 2; do
    python bench.py --no-cpu-baseline --no-extra --content $c --resampler fir --option firsync=$fs --steps 20 --warmup 10 > $o/fir_${c}_sync$fs.json 2> $o/err.txt
  done
  python bench.py --no-cpu-baseline --no-extra --content $c --steps 20 --warmup 10 > $o/box_$c.json 2>> $o/err.txt
done
python - $o <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j = json.load(open(f)); r = j["roofline"]
    print(f"{os.path.basename(f):28s} {j['value']:10.1f} kernel {r['kernel_ms_per_step']:.4f} ms frac {r['frac']:.4f} {r['variant']} redone {j.get('frames_redone')}")
PY
