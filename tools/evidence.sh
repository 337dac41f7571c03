#!/bin/bash
# Regenerate the measurement bundle of a round on the GPU box (run through gpurun): bench lines, rocprofv3 kernel trace
# (+ stats), PMC passes, per-layer conv logs, the other BASELINE configs.  Output under gpurun_out/<name>/.
OUT=${1:-gpurun_out/evidence}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/$OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 --api composed --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_composed.json
IVX_NARROW_EPILOGUE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_narrow_epilogue.json
python bench.py --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bf16.json
IVX_WINOGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_direct.json
IVX_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_dist1.json
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1 lift_nuscenes lift_scannet; do python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl; done
for c in scannet_v1 scannet_fast sunrgbd_fast; do python bench.py --config $c --storage bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl; done
for c in scannet_v1 scannet_fast; do python bench.py --config $c --storage bf16 --trunk-fp8 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl; done
for c in scannet_v1 scannet_fast; do python bench.py --config $c --storage bf16 --trunk-fp8 --fp8-residual fp8 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl; done
python bench.py --config scannet_fast --views 20 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl
python tools/wino_ab.py > $OUT/wino_ab.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace_bench.log 2>&1)
grep '^{"metric' $OUT/trace_bench.log | tail -1 > $OUT/bench_profiled.json
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 7 > $OUT/kernel_trace.md
find $OUT/trace -name "*stats*.csv" | head -3 | while read f; do cp $f $OUT/; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_scannet -o t -- python $ROOT/bench.py --config scannet_fast --steps 3 --warmup 1 > $ROOT/$OUT/trace_scannet.log 2>&1)
DB=$(find $OUT/trace_scannet -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_trace_scannet_fast.md
grep '^{"metric' $OUT/trace_scannet.log | tail -1 > $OUT/bench_profiled_scannet_fast.json
python tools/trunk_layers.py --config scannet_v1 --dtype f32 --top 40 > $OUT/trunk_layers_scannet_v1_f32.md 2>/dev/null
python tools/trunk_layers.py --config kitti --top 40 > $OUT/trunk_layers_kitti.md 2>/dev/null
bash tools/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc --min-ms 0.5 --json $OUT/pmc.json > $OUT/pmc.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.1 --match wino_ --json $OUT/pmc_wino.json > $OUT/pmc_wino.md
rm -rf $OUT/trace/*/*.db.bak $OUT/trace $OUT/trace_scannet $OUT/trace_v1bf16 2>/dev/null
du -sh $OUT
