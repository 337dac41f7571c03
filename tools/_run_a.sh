#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline_winograd_transforms']['ms_per_step'], r['roofline_winograd_transforms']['achieved'])"
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd" 2>&1 | tail -3
