cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x7
timeout 900 python -m pytest tests/test_gpu_feeder.py -x -q > gpurun_out/x7/test.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/x7/test.log
timeout 600 python tools/feeder_bench.py > gpurun_out/x7/feeder.log 2>&1; tail -5 gpurun_out/x7/feeder.log
