cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x6
timeout 900 python -m pytest tests/test_criterion_golden.py tests/test_training_step.py -x -q -m gpu > gpurun_out/x6/test.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/x6/test.log
