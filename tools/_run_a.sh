#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fp8" -s 2>&1 | grep -E "fp8 trunk|passed|failed|Error" | head
for c in scannet_v1 scannet_fast; do
for f in "" "--trunk-fp8"; do
timeout 300 python bench.py --config $c --storage bf16 $f --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/b.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); t=r.get('roofline_trunk_2d') or {}
print(r['config']['workload'], r['dtype'], r['value'], r['ms_per_step'], 'neck', r['roofline']['neck_ms_per_step'], 'trunk', t.get('ms_per_step'), t.get('achieved'), 'dets', r['config']['detections_last_step'])" || tail -5 gpurun_out/b.err
done; done
