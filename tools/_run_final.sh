cd $GRAFT_REPO_ROOT
O=gpurun_out/r2final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
( time python bench.py ) > $O/bench_noflags.json 2> $O/bench_noflags.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-700 $O/bench_driver.json; grep real $O/bench_noflags.err
