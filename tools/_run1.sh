cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "5shot" 2>&1 | tail -25 | cut -c1-200
