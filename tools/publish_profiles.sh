#!/bin/bash
# tools/publish_profiles.sh ROUND -- copy what is to be judged from gpurun_out/profiles_ROUND/ (scratch) into profiles/ (tracked)
set -e
R=${1:-r03}
S=gpurun_out/profiles_$R
cp $S/bench.json profiles/${R}_bench.json
cp $S/traffic.json profiles/${R}_pmc_traffic.json
for n in c2box c2fir c3 c4 c4fir c1; do
  [ -f $S/summary_$n.txt ] || continue
  cp $S/summary_$n.txt profiles/${R}_${n}_pmc_summary.txt
  f=$(ls -t $S/$n/trace/*/*kernel_stats.csv | head -1)  # the newest: gpurun merges into what earlier calls left
  cp "$f" profiles/${R}_${n}_kernel_stats.csv
done
for f in tfbench.txt dense.txt streambench.txt bench_c5_single_gpu.json content.txt layout.txt groups.txt firsync.txt fir_floor.txt blocktimes.txt tail_ab.txt; do [ -f $S/$f ] && cp $S/$f profiles/${R}_$f; done
ls -la profiles/ | grep $R
