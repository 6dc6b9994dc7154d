bash tools/bench_trace.sh bt1 4 | head -8
python tools/summarize_trace.py $(ls gpurun_out/bt1/prof/*/*kernel_trace.csv) k_conv > gpurun_out/bt1/conv_by_level.md
rm -f gpurun_out/bt1/prof/*/*kernel_trace.csv
