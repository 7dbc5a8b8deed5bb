for r in 1 2 3; do
 for f in 16 32 64 128; do
  python bench.py --no-extra --no-cpu-baseline --resampler fir --frames $f --steps $((1280/f)) --warmup 10 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); r=j['roofline']; print('frames $f', j['value'], r['kernel_ms_per_step'], r['frac'], r['variant'])"
 done
done
