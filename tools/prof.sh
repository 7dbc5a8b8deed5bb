#!/bin/bash
# tools/prof.sh OUTDIR -- CMD...   (GPU box)  kernel-trace stats + PMC passes for CMD.
# PMC passes run separately from the trace pass (gpurun refuses combined tracing+pmc).
set -u
OUT=$1; shift; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$@" > "$OUT/trace.log" 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- "$@" > "$OUT/pmc$i.log" 2>&1
done
find "$OUT" -name "*.csv" | head -30
