cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_pair_chain.py tests/test_gpu_pair.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-cabi 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline_trunk_2d']['ms_per_step'], d['roofline']['neck_ms_per_step'], d['exact_fp32_mfma']['value'])
for e in d.get('extra_configs', []): print(e['workload'][:60], e['value'], e['ms_per_step'])"
