#!/bin/bash
# rebuild helper (ignored by git): whatever the caller's cwd
cd /root/repo && python -m imvoxelnet_amd._build | tail -1 && python oracle/cpu_abi/build.py | tail -1
