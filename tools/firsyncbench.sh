# tools/firsyncbench.sh -- k_fir_fused with its blocks' waves meeting at a barrier every 0 (never), 1, 2, 4, 8 steps: C2 FIR and C4 FIR, rounds alternating
for r in 1 2 3 4; do
  for p in 0 1 2 4 8; do
    python bench.py --no-extra --no-cpu-baseline --resampler fir --option firsync=$p 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('C2 firsync $p', j['roofline']['kernel_ms_per_step'], j['verified'])"
    python bench.py --no-extra --no-cpu-baseline --resampler fir --workload C4 --frames 16 --option firsync=$p 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('C4 firsync $p', j['roofline']['kernel_ms_per_step'], j['verified'])"
  done
done
