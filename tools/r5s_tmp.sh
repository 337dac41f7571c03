cd /root/repo
mkdir -p gpurun_out/r5u
for r in 1 2; do
PIO_CFGS=74,174,194,66,166,186 timeout 300 python tools/pio_scaling.py --iters 100 > gpurun_out/r5u/scaling$r.txt 2>&1
grep -A9 "K sweep\|tile-count" gpurun_out/r5u/scaling$r.txt | head -40
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-cabi"
for r in 1 2 3; do
for v in 100 120; do
  IVX_BENCH_EXTRA=0 IVX_PIO_DEEP_ADD=$v timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('add $v', d['value'], d['ms_per_step'], d['roofline_trunk_2d']['ms_per_step'])"
done
done
