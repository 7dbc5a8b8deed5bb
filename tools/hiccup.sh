# tools/hiccup.sh -- the full bench line several times: does a secondary pass hold a step many times longer than the others?
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print($i, j['roofline']['frac'], ' '.join(f\"{k}={v['frac']}\" for k, v in j['others'].items()))
for k, v in j['others'].items():
    if v.get('kernel_ms_steps'): print('   ', k, v['kernel_ms_steps'])"
done
