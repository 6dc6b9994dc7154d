R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r2h
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/gputest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r2h/gputest.log
timeout 600 python bench.py --steps 30 --warmup 8 > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc $?"
cut -c1-300 gpurun_out/r2h/bench.json
bash tools/bench_trace.sh r2h_trace 30
timeout 900 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r2h/configs.txt
