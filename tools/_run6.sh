mkdir -p gpurun_out/c1
python -m pytest tests -m gpu -q -x > gpurun_out/c1/gputest.txt 2>&1; tail -3 gpurun_out/c1/gputest.txt
python bench.py --steps 30 --warmup 8 > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; cut -c1-200 gpurun_out/c1/bench.json
bash tools/bench_trace.sh c1t 6 | tail -8
python tools/summarize_trace.py $(ls gpurun_out/c1t/prof/*/*kernel_trace.csv) k_conv > gpurun_out/c1/conv_by_level.md
python tools/trace_span.py $(ls gpurun_out/c1t/prof/*/*kernel_trace.csv) k_voxelize_fp 25 6 > gpurun_out/c1/forward_timeline.txt 2>&1
cp gpurun_out/c1t/prof/*/*kernel_stats.csv gpurun_out/c1/bench_kernel_stats.csv
cp gpurun_out/c1t/rocprof_conv_l1.json gpurun_out/c1/
rm -rf gpurun_out/c1t/prof
