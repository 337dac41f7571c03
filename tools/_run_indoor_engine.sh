#!/bin/bash
# gpurun helper: the indoor native-handle tests + the whole engine file
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/indoor_engine.log
cat gpurun_out/indoor_engine.log
