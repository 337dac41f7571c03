#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "bf16" 2>&1 | tail -15
for c in scannet_v1 scannet_fast sunrgbd_fast; do
for s in bf16 f32; do
timeout 300 python bench.py --config $c --storage $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); t=r.get('roofline_trunk_2d') or {}
print(r['config']['workload'], r['dtype'], r['value'], r['ms_per_step'], 'neck', r['roofline']['neck_ms_per_step'], 'trunk', t.get('ms_per_step'), t.get('achieved'))"
done; done
timeout 300 python bench.py --storage bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
