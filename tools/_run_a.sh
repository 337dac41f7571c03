#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/trunk_layers.py --config scannet_v1 --dtype bf16 --top 40 2>/dev/null > gpurun_out/trunk_layers_v1_bf16.md
cat gpurun_out/trunk_layers_v1_bf16.md | head -34
