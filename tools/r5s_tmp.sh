cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pair_chain.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/conv_timeline_model.py --shapes '64,256,1,96;128,512,1,48;256,1024,1,24;512,2048,1,12' 2>&1 | grep -v amdgpu.ids
for r in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cpu-cabi 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline_trunk_2d']['ms_per_step'], d['roofline']['neck_ms_per_step'])
for e in d.get('extra_configs', []): print(e['workload'][:60], e['value'], e['ms_per_step'])"
done
