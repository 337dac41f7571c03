#!/bin/bash
# Round-3 (second half: pair operands) measurement bundle on the GPU box (run through gpurun).  Output under gpurun_out/<name>/.
OUT=${1:-gpurun_out/evidence_r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/$OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 --wino-operands f32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_f32_operands.json
if [ -z "$IVX_EVIDENCE_SHORT" ]; then      # IVX_EVIDENCE_SHORT=1: only the default / fp32-operand lines, the kernel trace and the PMC passes
python bench.py --steps 20 --warmup 5 --api composed --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_composed.json
python bench.py --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bf16.json
IVX_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_dist1.json
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1; do python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl; done
python bench.py --config scannet_fast --views 20 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1; do IVX_WINO_OPERANDS=0 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_f32_operands.jsonl; done
python tools/pair_ab.py --reps 5 --cfgs 0,75,76,81,82,85 > $OUT/pair_ab.log 2>&1
fi
(cd /tmp && IVX_BENCH_ALT=0 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace_bench.log 2>&1)
grep '^{"metric' $OUT/trace_bench.log | tail -1 > $OUT/bench_profiled.json
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 7 > $OUT/kernel_trace.md
find $OUT/trace -name "*stats*.csv" | head -3 | while read f; do cp $f $OUT/; done
[ -z "$IVX_EVIDENCE_SHORT" ] && python tools/trunk_layers.py --config kitti --top 40 > $OUT/trunk_layers_kitti.md 2>/dev/null
bash tools/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc --min-ms 0.3 --json $OUT/pmc.json > $OUT/pmc.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.1 --match wino_ --json $OUT/pmc_wino.json > $OUT/pmc_wino.md
rm -rf $OUT/trace $OUT/pmc/pass*/*.db 2>/dev/null
du -sh $OUT
