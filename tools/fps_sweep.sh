for g in 4 6 9 12 16; do echo "G=$g"; GF_FPS_G=$g python tools/bench_points.py 2>&1 | grep -E "^fps"; done
