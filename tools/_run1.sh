cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x15
timeout 900 python -m pytest tests/test_gpu_geodesic.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
for fb in 0 11; do echo "filter bits $fb"; GF_BFS_FILTER_BITS=$fb timeout 300 python tools/bench_bfs_sources.py 2>&1 | grep "fps picks" | grep "1239\|1234"; done
for fb in 0 11 12; do GF_BFS_FILTER_BITS=$fb timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x15/b.log 2>&1; echo "filter bits $fb rc $?: $(grep '^{' gpurun_out/x15/b.log | cut -c1-100)"; done
