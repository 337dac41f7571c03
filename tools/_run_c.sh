cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py "tests/test_gpu_model.py::test_kitti_full_path_vs_oracle" tests/test_gpu_model.py::test_graphed_simple_test_equals_eager tests/test_gpu_kernels.py::test_e2e_small_golden -m gpu -q -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
for st in 0 50 100 200; do python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 0,1,2,4,5 --stagger $st 2>&1 | grep winograd | sed "s/^/stagger $st: /"; done > $O/stagger.log
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 2,3,4 --wcfgs 54,55,51 2>&1 | grep winograd > $O/wcfgs.log
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 0,1 --wcfgs 56,49,53,57 2>&1 | grep winograd >> $O/wcfgs.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python bench.py --steps 10 --warmup 3 --api composed --no-cpu-baseline > $O/bench_composed.json 2>> $O/bench.err
IVX_NATIVE_MODEL=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pyhost.json 2>> $O/bench.err
tail -25 $O/pytest.log | cut -c1-300; tail -4 $O/smoke.log; cat $O/stagger.log | cut -c1-200; cat $O/wcfgs.log | cut -c1-200; cut -c1-400 $O/bench.json; tail -5 $O/bench.err
