cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pair_chain.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-cabi"
for r in 1 2; do
for v in "512 512" "1024 1024" "2048 2048"; do set -- $v
  IVX_BENCH_EXTRA=0 IVX_PIO_DEEP_TILES=$1 IVX_PIO_W8_TILES=$2 timeout 300 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('deep_tiles $1 w8 $2', d['value'], d['ms_per_step'], d['roofline_trunk_2d']['ms_per_step'])"
done
done
