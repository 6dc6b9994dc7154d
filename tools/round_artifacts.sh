# Dev tool: the round's artifacts in one GPU call: GPU test suite, bench line, rocprofv3 kernel stats of bench.py, per-level
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
# conv table, forward timeline -> gpurun_out/<tag>/   (copy what is to be judged into profiles/)
tag=${1:-art}; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/$tag/gputest.txt 2>&1; tail -3 gpurun_out/$tag/gputest.txt
timeout 600 python bench.py --steps 30 --warmup 8 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; cut -c1-200 gpurun_out/$tag/bench.json
bash tools/bench_trace.sh ${tag}_t 6 | tail -8
python tools/summarize_trace.py $(ls gpurun_out/${tag}_t/prof/*/*kernel_trace.csv) k_conv > gpurun_out/$tag/conv_by_level.md
python tools/trace_span.py $(ls gpurun_out/${tag}_t/prof/*/*kernel_trace.csv) k_voxelize_fp 25 10 > gpurun_out/$tag/forward_timeline.txt 2>&1
python tools/trace_span.py $(ls gpurun_out/${tag}_t/prof/*/*kernel_trace.csv) k_voxelize_fp 0 10 > gpurun_out/$tag/forward_all_kernels.txt 2>&1
cp gpurun_out/${tag}_t/prof/*/*kernel_stats.csv gpurun_out/$tag/bench_kernel_stats.csv
cp gpurun_out/${tag}_t/rocprof_conv_l1.json gpurun_out/${tag}_t/rocprof_conv_family.json gpurun_out/$tag/
rm -rf gpurun_out/${tag}_t/prof
