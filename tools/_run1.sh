cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x8
for r in 1 2 3; do
timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x8/b_$r.log 2>&1; echo "rc $?: $(grep '^{' gpurun_out/x8/b_$r.log | cut -c1-120)"
done
