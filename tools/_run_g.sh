cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py -q -m gpu -k "native or graphed or kitti_full or c_program" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_graph_$i.json 2>> $O/bench.err
IVX_NATIVE_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_eager_$i.json 2>> $O/bench.err
done
tail -3 $O/pytest.log; tail -1 $O/smoke.log; for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
r=json.loads(open('$f').read()); ro=r['roofline']
print(r['value'], r['ms_per_step'], ro['achieved'], ro['frac'], ro['neck_ms_per_step'], ro['launches_per_step'], r['config']['device_side'])"; done; tail -3 $O/bench.err
