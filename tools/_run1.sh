cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x3
timeout 600 python -m pytest tests/test_gpu_unet_exec.py -x -q > gpurun_out/x3/test.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/x3/test.log
python tools/host_vs_gpu.py > gpurun_out/x3/host_vs_gpu.log 2>&1
tail -5 gpurun_out/x3/host_vs_gpu.log
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/bbt; mkdir -p $R/gpurun_out/bbt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bbt/prof -- python3 $R/tools/backbone_trace.py > $R/gpurun_out/bbt/log 2>&1
python3 $R/tools/trace_span.py $(ls $R/gpurun_out/bbt/prof/*/*kernel_trace.csv) k_voxelize_fp 0 > $R/gpurun_out/bbt/span.txt
tail -1 $R/gpurun_out/bbt/span.txt
