#!/bin/bash
# PMC passes on one neck layer (tools/conv_bench.py --layers $1 --cfgs $2).  Counters are collected in their own
# runs (no trace domains besides kernel dispatch), one hardware-counter group per pass.
LAYER=${1:-2}; CFG=${2:-0}; OUT=${3:-gpurun_out/pmc}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/$OUT/pass$i -o p -- \
      python $ROOT/tools/conv_bench.py --layers $LAYER --cfgs $CFG --iters 2 > $ROOT/$OUT/pass$i.log 2>&1
done
find $ROOT/$OUT -name "*counter_collection.csv" | head
