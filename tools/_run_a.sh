cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python tools/overlap_ubench.py --part all > gpurun_out/r2a/overlap.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_model.py::test_kitti_bf16_storage_mode_tracks_fp32 tests/test_gpu_configs.py tests/test_gpu_kernels.py::test_pipelined_stack_equals_sequential tests/test_gpu_kernels.py::test_unprojection_crop_larger_than_map_is_clamped tests/test_gpu_model.py -s > gpurun_out/r2a/pytest_new.log 2>&1
echo "pytest rc $?" >> gpurun_out/r2a/pytest_new.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
IVX_PIPE_CHUNKS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_seq.json 2>> gpurun_out/r2a/bench.err
tail -30 gpurun_out/r2a/overlap.log; tail -15 gpurun_out/r2a/pytest_new.log; cat gpurun_out/r2a/bench.json | cut -c1-400
