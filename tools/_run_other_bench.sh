#!/bin/bash
# gpurun helper: bench lines of the other configs through the public call (native handle; traced / untraced) and composed
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/other_bench.jsonl
: > $out
for rep in 1 2; do
for c in scannet_fast sunrgbd_fast scannet_v1; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>gpurun_out/other_bench.err | tail -1 >> $out
  IVX_BENCH_TRACE=0 timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>gpurun_out/other_bench.err | tail -1 >> $out
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --api composed 2>>gpurun_out/other_bench.err | tail -1 >> $out
done
done
python - <<'PY'
import json
for l in open('gpurun_out/other_bench.jsonl'):
    try:
        r = json.loads(l)
    except Exception:
        print('BAD', l[:200]); continue
    print(r['config']['workload'], r['config'].get('api'), r['value'], r['scenes_per_s'], r['ms_per_step'], r['roofline']['neck_ms_per_step'], r['roofline']['achieved'],
          (r.get('roofline_trunk_2d') or {}).get('ms_per_step'), (r.get('roofline_trunk_2d') or {}).get('achieved'))
PY
tail -5 gpurun_out/other_bench.err
