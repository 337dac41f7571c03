#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/fp8_noise.py 2>&1 | grep -v amdgpu.ids | tail -8
for st in 4 3 2 1; do
timeout 300 python bench.py --config scannet_v1 --storage bf16 --trunk-fp8 --fp8-stages $st --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/b.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); t=r.get('roofline_trunk_2d') or {}
print('stages $st', r['value'], r['ms_per_step'], 'trunk', t.get('ms_per_step'))" || tail -5 gpurun_out/b.err
done
