# Dev tool: kernel trace of a short bench run + timeline of its last forward.  usage: forward_timeline.sh <tag> [ENV=val ...]
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
R=$GRAFT_REPO_ROOT; tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/forward_timeline.py $(ls $R/gpurun_out/tl_$tag/*/*kernel_trace.csv | head -1) 100 20
