cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x14
timeout 900 python -m pytest tests/test_gpu_geodesic.py tests/test_gpu_fullsize.py -x -q -k "bfs or geodesic" 2>&1 | tail -2
timeout 300 python tools/bench_bfs.py 2>&1 | tail -13
for q in 1 2 4; do GF_BFS_QPW=$q timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x14/b.log 2>&1; echo "qpw $q rc $?: $(grep '^{' gpurun_out/x14/b.log | cut -c1-100)"; done
