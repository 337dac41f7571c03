cd /root/repo
ROOT=/root/repo; OUT=gpurun_out/evidence_r05; mkdir -p $OUT
export TMPDIR=/tmp
trace() {   # name, steps-profiled, bench args...
  name=$1; nst=$2; shift 2
  (cd /tmp && IVX_BENCH_ALT=0 IVX_BENCH_EXTRA=0 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_$name -o t -- python $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/$OUT/trace_$name.log 2>&1)
  grep '^{"metric' $OUT/trace_$name.log | tail -1 > $OUT/bench_profiled_$name.json
  DB=$(find $OUT/trace_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $nst > $OUT/kernel_trace_$name.md
  rm -rf $OUT/trace_$name
}
trace kitti 7 --steps 5 --warmup 2
rm -rf $OUT/pmc
bash tools/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $OUT/pmc --min-ms 0.3 --json $OUT/pmc.json > $OUT/pmc.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.1 --match wino_ --json $OUT/pmc_wino.json > $OUT/pmc_wino.md
python tools/pmc_summary.py $OUT/pmc --min-ms 0.02 --match "conv_igemm_v4_kernel<__bf16" --json $OUT/pmc_trunk.json > $OUT/pmc_trunk.md
rm -rf $OUT/pmc*/pass*/*.db 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
head -30 $OUT/kernel_trace_kitti.md | cut -c1-160
