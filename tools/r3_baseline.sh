# tools/r3_baseline.sh OUTDIR -- round-3 starting point: block finish times (box, FIR), content, transfer pairs, inverse
o=${1:-gpurun_out/r3b}; mkdir -p $o
H2Y_BLOCK_TIMES_FILE=$o/bt_box python bench.py --no-extra --no-cpu-baseline --lib build/exp_bt/libh2y_bt.so --allow-experiment --steps 12 --warmup 10 > $o/bt_box.json 2> $o/bt_box.err
H2Y_BLOCK_TIMES_FILE=$o/bt_fir python bench.py --no-extra --no-cpu-baseline --resampler fir --lib build/exp_bt/libh2y_bt.so --allow-experiment --steps 12 --warmup 10 > $o/bt_fir.json 2> $o/bt_fir.err
for k in 16 17 18 19 20 21; do python tools/blocktimes.py $o/bt_box.$k; done > $o/bt_box.txt 2>&1
python tools/blockcorr.py $o/bt_box.18 $o/bt_box.19 $o/bt_box.20 $o/bt_box.21 >> $o/bt_box.txt 2>&1
for k in 16 17 18 19 20 21; do python tools/blocktimes.py $o/bt_fir.$k; done > $o/bt_fir.txt 2>&1
bash tools/contentbench.sh $o/content > $o/content.txt 2>&1
python tools/tfbench.py > $o/tfbench.txt 2>&1
python tools/kbench.py inv u16 pass > $o/kbench.txt 2>&1
rm -f $o/bt_box.[0-9]* $o/bt_fir.[0-9]*
