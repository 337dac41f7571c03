cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
python tools/_diag_engine.py > $O/diag.log 2>&1
cat $O/diag.log | grep -v amdgpu
