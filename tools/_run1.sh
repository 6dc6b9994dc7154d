cd $GRAFT_REPO_ROOT
GF_LIB_PATH=geoformer_amd/lib/variants/fpstrace.so timeout 300 python tools/trace_fps.py 2>&1 | tail -34
