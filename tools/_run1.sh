cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_unet_exec.py -x -q 2>&1 | tail -12 | cut -c1-200
