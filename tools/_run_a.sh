#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in f32 bf16; do
  bash tools/pmc_bench.sh gpurun_out/pmc_v1_$v --config scannet_v1 --storage $v > gpurun_out/pmc_v1_$v.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_v1_$v --min-ms 0.05 > gpurun_out/pmc_v1_$v.md
  rm -rf gpurun_out/pmc_v1_$v/pass*/*.db 2>/dev/null
done
bash tools/pmc_bench.sh gpurun_out/pmc_v1_fp8 --config scannet_v1 --storage bf16 --trunk-fp8 > gpurun_out/pmc_v1_fp8.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_v1_fp8 --min-ms 0.05 > gpurun_out/pmc_v1_fp8.md
du -sh gpurun_out/pmc_v1_*; grep -c "^## " gpurun_out/pmc_v1_f32.md gpurun_out/pmc_v1_bf16.md gpurun_out/pmc_v1_fp8.md
rm -rf gpurun_out/pmc_v1_f32 gpurun_out/pmc_v1_bf16 gpurun_out/pmc_v1_fp8
