#!/bin/bash
# tools/r3_experiments.sh OUTDIR (GPU box; needs build/exp_nc1, exp_nc2, exp_bt: -DH2Y_EXPERIMENT builds, see profiles/README.md)
#   fir_floor.txt    k_fir_fused and k_fused_t1 as built against the timing builds that leave the pixel arithmetic out (NOCOMPUTE=1)
#                    and the FIR stages too (=2): what the kernels' loads, stores, tickets and barriers take by themselves
#   blocktimes.txt   when the blocks of a 64 x 4K launch finish (box kernel with the dynamic last frame off / on; FIR kernel)
#   tail_ab.txt      the dynamic last frame off against on, and slices by block speed against by XCD speed: ABBA, six runs each
o=${1:-gpurun_out/r3x}; mkdir -p $o
one() { python bench.py --no-extra --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms_per_step'], j['roofline']['frac'], j['roofline']['variant'], 'experiment' if j.get('experiment_build') else 'product')"; }
{
  for r in 1 2 3; do
    echo "C2 FIR  product        $(one --resampler fir)"
    echo "C2 FIR  no pixel math  $(one --resampler fir --lib build/exp_nc1/libh2y_nc1.so --allow-experiment)"
    echo "C2 FIR  + no FIR math  $(one --resampler fir --lib build/exp_nc2/libh2y_nc2.so --allow-experiment)"
    echo "C2 box  product        $(one)"
    echo "C2 box  no pixel math  $(one --lib build/exp_nc1/libh2y_nc1.so --allow-experiment)"
  done
} > $o/fir_floor.txt 2>&1
{
  for v in "tail=off" "tail=auto"; do
    H2Y_BLOCK_TIMES_FILE=$o/bt python bench.py --no-extra --no-cpu-baseline --no-pipeline --lib build/exp_bt/libh2y_bt.so --allow-experiment --steps 6 --warmup 10 --option $v > /dev/null 2>&1
    for k in 13 14 15; do echo "== box kernel, $v, launch $k"; python tools/blocktimes.py $o/bt.$k; done
    rm -f $o/bt.*
  done
  H2Y_BLOCK_TIMES_FILE=$o/bt python bench.py --no-extra --no-cpu-baseline --no-pipeline --resampler fir --lib build/exp_bt/libh2y_bt.so --allow-experiment --steps 6 --warmup 10 > /dev/null 2>&1
  for k in 14 15; do echo "== FIR kernel, launch $k"; python tools/blocktimes.py $o/bt.$k; done
  rm -f $o/bt.*
} > $o/blocktimes.txt 2>&1
{
  echo "# A = --option tail=off, B = --option tail=auto (the dynamic last frame), C2 box"
  bash tools/ab_opt.sh $o/ab_tail "--option tail=off" "--option tail=auto"
  echo "# A = --option balance=xcd, B = default (slices by block speed), C2 box"
  bash tools/ab_opt.sh $o/ab_bal "--option balance=xcd" ""
} > $o/tail_ab.txt 2>&1
rm -rf $o/ab_tail $o/ab_bal
