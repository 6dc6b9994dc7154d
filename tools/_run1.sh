cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x8
for q in 4 8 16; do for r in 1 2; do
GPU_MAX_HW_QUEUES=$q GF_BENCH_STREAMS=$r timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x8/b_$r.log 2>&1; echo "queues $q streams $r rc $?: $(grep '^{' gpurun_out/x8/b_$r.log | cut -c1-100)"
done; done
