R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/ft3; mkdir -p $R/gpurun_out/ft3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ft3/prof -- python3 $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-secondary > $R/gpurun_out/ft3/log 2>&1
python3 $R/tools/summarize_trace.py $(ls $R/gpurun_out/ft3/prof/*/*kernel_trace.csv) k_conv_g16p
