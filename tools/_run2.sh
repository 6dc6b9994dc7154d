python tools/ab_flat.py 50 2>&1 | tail -1
python tools/nq128_check.py 2>&1 | grep "ms per scene"
GPU_MAX_HW_QUEUES=4 python tools/nq128_check.py 2>&1 | grep "ms per scene"
GF_BFS_WG=512 python tools/nq128_check.py 2>&1 | grep "ms per scene"
