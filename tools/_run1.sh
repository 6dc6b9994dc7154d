R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/ft4; mkdir -p $R/gpurun_out/ft4
GF_FPS_FIRST=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ft4/prof -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-secondary > $R/gpurun_out/ft4/log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/ft4/prof/*/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:],r['Stream_Id']) for r in csv.DictReader(open(f))]
rows.sort()
T0=None
for s,e,n,st in rows:
    if 'k_fps' in n or 'k_geodesic_bfs' in n or 'k_stream_gate' in n or 'k_decoder_cross_attn' in n:
        if T0 is None: T0=s
        print("%10.1f %8.1f s%-3s %s"%((s-T0)/1e3,(e-s)/1e3,st,n))
PY
