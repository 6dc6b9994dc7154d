for v in 0 1; do echo LDSW=$v; GF_CONV_LDSW=$v python tools/bench_conv.py 2>&1 | grep -E "^level [12] |L1 both"; done
