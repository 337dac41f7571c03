cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
echo "pytest rc $?" >> $O/pytest_all.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python bench.py --steps 10 --warmup 3 --api composed --no-cpu-baseline > $O/bench_composed.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --graph --no-cpu-baseline > $O/bench_graph.json 2>> $O/bench.err
tail -15 $O/pytest_all.log | cut -c1-250; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
