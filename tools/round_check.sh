R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r2g
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2g/gputest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r2g/gputest.log
timeout 600 python bench.py --steps 30 --warmup 8 > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc $?"
cut -c1-300 gpurun_out/r2g/bench.json
bash tools/bench_trace.sh r2g_trace 30
