for w in 10 400 10 100 10 1000 10; do
  python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup $w 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('warmup $w', j['value'], j['roofline']['kernel_ms_per_step'])"
done
