#!/bin/bash
# Round-6 measurement bundle on the GPU box (run through gpurun).  Output under gpurun_out/<name>/; tools/evidence_to_profiles.py copies the
# summaries into profiles/r06_*.
OUT=${1:-gpurun_out/evidence_r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/$OUT
cd $ROOT
if [ "$EVIDENCE_PROFILES_ONLY" != "1" ]; then
nproc > $OUT/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/host.txt
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
NOX="env IVX_BENCH_EXTRA=0"
$NOX python bench.py --steps 10 --warmup 3 --wino-operands f32 --trunk-operands f32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_f32_operands.json
$NOX python bench.py --steps 10 --warmup 3 --trunk-operands f32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_trunk_f32.json
$NOX python bench.py --steps 10 --warmup 3 --api composed --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_composed.json
$NOX python bench.py --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bf16.json
# A/B of this round's one-launch kernels inside the whole step
IVX_FUSE_BOTTLENECK=0 $NOX python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_no_fused_bottleneck.json
IVX_FUSE_STEM=0 $NOX python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_no_fused_stem.json
IVX_FUSE_STEM=0 IVX_FUSE_BOTTLENECK=0 $NOX python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_no_fusion.json
IVX_BENCH_FORCE_DIST=1 $NOX python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_dist1.json
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1; do python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl; done
python bench.py --config scannet_fast --views 20 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other.jsonl
for c in nuscenes scannet_v1; do IVX_FUSE_STEM=0 IVX_FUSE_BOTTLENECK=0 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_no_fusion.jsonl; done
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1; do python bench.py --config $c --steps 10 --warmup 3 --wino-operands f32 --trunk-operands f32 2>/dev/null | tail -1 >> $OUT/other_f32_operands.jsonl; done
python bench.py --config scannet_v1 --steps 10 --warmup 3 --storage bf16 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl
python bench.py --config scannet_v1 --steps 10 --warmup 3 --storage bf16 --trunk-fp8 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl
python bench.py --config scannet_v1 --steps 10 --warmup 3 --storage bf16 --trunk-fp8 --fp8-variant full 2>/dev/null | tail -1 >> $OUT/other_bf16.jsonl
# A/B of the split-operand form of the non-Winograd 3x3x3 neck layers + K split of the narrow head convs (round 6, last change) inside the whole step
for c in nuscenes sunrgbd_fast scannet_fast scannet_v1; do IVX_CONV_PAIR=0 IVX_CONV_SKINNY=0 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT/other_no_split_form.jsonl; done
for c in scannet_v1 scannet_fast sunrgbd_fast nuscenes; do python tools/neck_layers.py --config $c --min-pos 256 > $OUT/neck_layers_$c.md 2>/dev/null; done
# the round-5 placement of the stage events (an event pair around every launch group of the timed steps)
IVX_BENCH_TRACE_TIMED=1 $NOX python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_events_in_timed_region.json
fi   # EVIDENCE_PROFILES_ONLY
# kernel traces (IVX_BENCH_TRACE_TIMED=1: the profiled process runs exactly warmup + steps model steps, no traced steps after the timed region)
trace() {   # name, steps-profiled, bench args...
  name=$1; nst=$2; shift 2
  (cd /tmp && IVX_BENCH_ALT=0 IVX_BENCH_EXTRA=0 IVX_BENCH_TRACE_TIMED=1 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_$name -o t -- python $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/$OUT/trace_$name.log 2>&1)
  grep '^{"metric' $OUT/trace_$name.log | tail -1 > $OUT/bench_profiled_$name.json
  DB=$(find $OUT/trace_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $nst > $OUT/kernel_trace_$name.md
  rm -rf $OUT/trace_$name
}
trace kitti 7 --steps 5 --warmup 2
trace scannet_v1 8 --config scannet_v1 --steps 4 --warmup 2
trace nuscenes 8 --config nuscenes --steps 4 --warmup 2
python tools/trunk_layers.py --config kitti --top 60 > $OUT/trunk_layers_kitti.md 2>/dev/null
python tools/trunk_layers.py --config scannet_v1 --top 60 > $OUT/trunk_layers_scannet_v1.md 2>/dev/null
# the one-launch kernels alone
python tools/bottleneck_ab.py --md $OUT/bottleneck_ab.md > /dev/null 2>&1
python tools/stem_ab.py --md $OUT/stem_ab.md > /dev/null 2>&1
# PMC passes (counters only)
bash tools/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc --min-ms 0.3 --json $OUT/pmc.json > $OUT/pmc.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.1 --match wino_ --json $OUT/pmc_wino.json > $OUT/pmc_wino.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.02 --match "conv_igemm_v4_kernel<__bf16,bottleneck_pio_kernel,stem_pool_pair_kernel" --json $OUT/pmc_trunk.json > $OUT/pmc_trunk.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.0 --match "" --steps 3 --json $OUT/pmc_all.json > $OUT/pmc_all.md
bash tools/pmc_bench.sh $OUT/pmc_scannet_v1 --config scannet_v1 > $OUT/pmc_scannet_v1.log 2>&1
python tools/pmc_summary.py $OUT/pmc_scannet_v1 --min-ms 0.1 --match conv_igemm,conv_wino_halo,conv_wino_zblk,wino_,backproject,bottleneck_pio,stem_pool --json $OUT/pmc_scannet_v1.json > $OUT/pmc_scannet_v1.md
rm -rf $OUT/pmc*/pass*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
