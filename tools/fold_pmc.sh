#!/bin/bash
# PMC passes over the fused GEMM + output launch of F(4x4,3x3) on the KITTI 64-channel layer, one-wave (IVX_FOLD4_WAVES=1) and two-wave (=2) kernels.
OUT=${1:-gpurun_out/fold_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/$OUT
cd /tmp
for wv in 1 2; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" \
             "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/$OUT/w$wv/pass$i -o p -- \
        env IVX_FOLD4_WAVES=$wv python $ROOT/tools/fused_ab.py --layers 64 --reps 1 > $ROOT/$OUT/w$wv.pass$i.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $ROOT/$OUT/w$wv --min-ms 0.3 --match fold4 > $ROOT/$OUT/fold_w$wv.md
done
find $ROOT/$OUT -name "*.csv" -size +2M -delete; rm -rf $ROOT/$OUT/w*/pass*/*.db
