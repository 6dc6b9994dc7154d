# Dev tool: GPU tests + bench + traced bench of the current tree -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; tag=${1:-r3}; mkdir -p $R/gpurun_out/$tag
cd $R
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$tag/gputest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/$tag/gputest.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k arbiter 2>&1 | grep -A14 "^arbiter" > gpurun_out/$tag/arbiter.txt; cat gpurun_out/$tag/arbiter.txt
timeout 900 python bench.py --steps 30 --warmup 8 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc $?"
tail -3 gpurun_out/$tag/bench.err
cut -c1-300 gpurun_out/$tag/bench.json
bash tools/bench_trace.sh ${tag}_trace 30
