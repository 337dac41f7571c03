cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/native_trace_$i.json 2>> $O/bench.err
IVX_BENCH_TRACE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/native_notrace_$i.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 --api composed --no-cpu-baseline > $O/composed_$i.json 2>> $O/bench.err
done
python bench.py --steps 10 --warmup 3 --graph --no-cpu-baseline > $O/graph.json 2>> $O/bench.err
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 2,3,4,5 --wcfgs 0,58,59 2>&1 | grep winograd > $O/wcfgs8.log
python tools/conv_bench.py --winograd --tile 6 --iters 5 --layers 0,1 --wcfgs 0 2>&1 | grep winograd >> $O/wcfgs8.log
for f in $O/*.json; do echo $f; cut -c1-160 $f; done; cat $O/wcfgs8.log | cut -c1-200; tail -3 $O/bench.err
