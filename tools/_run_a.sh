#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "aligned or nms or topk" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_configs.py -x -q -m gpu -k "indoor or scannet or sunrgbd" 2>&1 | tail -5
ROOT=$GRAFT_REPO_ROOT; OUT=gpurun_out/bf16prof; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --config scannet_fast --storage bf16 --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 7 > $OUT/kernel_trace_scannet_fast_bf16.md
rm -rf $OUT/trace
grep -v "conv_igemm" $OUT/kernel_trace_scannet_fast_bf16.md | head -24 | cut -c1-150
