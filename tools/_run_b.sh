cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv or neck or pipelined or e2e or head" > $O/pytest_conv.log 2>&1
echo "pytest rc $?" >> $O/pytest_conv.log
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k scannet_v1 > $O/pytest_v1.log 2>&1
echo "pytest rc $?" >> $O/pytest_v1.log
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 0,1,2,3,4,5 > $O/wino_wide.log 2>&1
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 0,1,2,3,4,5 --narrow > $O/wino_narrow.log 2>&1
python tools/conv_bench.py --set resnet --iters 10 > $O/resnet_wide.log 2>&1
python tools/conv_bench.py --set resnet --iters 10 --narrow > $O/resnet_narrow.log 2>&1
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
IVX_NARROW_EPILOGUE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_narrow.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --api composed --no-cpu-baseline > $O/bench_composed.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --graph --no-cpu-baseline > $O/bench_graph.json 2>> $O/bench.err
tail -3 $O/pytest_conv.log; tail -12 $O/pytest_v1.log; grep -v amdgpu $O/wino_wide.log | grep winograd; grep -v amdgpu $O/wino_narrow.log | grep winograd; cut -c1-300 $O/bench.json; cut -c1-200 $O/bench_narrow.json
