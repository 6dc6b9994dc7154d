cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x2
timeout 600 python -m pytest tests/test_gpu_unet_exec.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q > gpurun_out/x2/test.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/x2/test.log
python tools/host_vs_gpu.py > gpurun_out/x2/host_vs_gpu.log 2>&1
tail -5 gpurun_out/x2/host_vs_gpu.log
