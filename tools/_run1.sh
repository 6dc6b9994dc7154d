cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "early_exit" 2>&1 | tail -25 | cut -c1-200
