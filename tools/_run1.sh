cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x5
for wg in 256 512 256 512; do
GF_BFS_WG=$wg timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x5/b_$wg.log 2>&1; echo "wg $wg rc $?: $(grep '^{' gpurun_out/x5/b_$wg.log | cut -c1-120)"; tail -2 gpurun_out/x5/b_$wg.log | cut -c1-300 | grep -v '^{'
done
