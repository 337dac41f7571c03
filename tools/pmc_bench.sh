#!/bin/bash
# PMC passes over the real bench command (counters only, one group per pass; never combined with trace domains
# other than kernel dispatch).  Output: gpurun_out/<name>/pass*/..._counter_collection.csv
# IVX_BENCH_TRACE_TIMED=1: no traced steps after the timed region, so the process runs exactly warmup + steps = 3 model steps (pmc_summary.py --steps 3).
OUT=${1:-gpurun_out/pmc_bench}
shift
ARGS="$@"     # extra bench.py arguments (e.g. --config scannet_v1 --storage bf16); default: the KITTI headline
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/$OUT/pass$i -o p -- \
      env IVX_BENCH_ALT=0 IVX_BENCH_EXTRA=0 IVX_BENCH_TRACE_TIMED=1 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > $ROOT/$OUT/pass$i.log 2>&1
  tail -1 $ROOT/$OUT/pass$i.log | cut -c1-200
done
