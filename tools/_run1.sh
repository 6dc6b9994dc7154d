cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x10
S=$(date +%s); timeout 1200 python -m pytest tests/test_training_step.py -x -q -m gpu -k full_size > gpurun_out/x10/test.log 2>&1; echo "rc $? in $(( $(date +%s) - S )) s"; tail -30 gpurun_out/x10/test.log | grep -v "^\s" | cut -c1-220 | tail -14
