R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for ov in 0 1; do
rm -rf $R/gpurun_out/ft5; mkdir -p $R/gpurun_out/ft5
GF_OVERLAP=$ov timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ft5/prof -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary > $R/gpurun_out/ft5/log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/ft5/prof/*/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f))]
rows.sort()
bfs=[round((e-s)/1e3) for s,e,n in rows if 'k_geodesic_bfs' in n]
fps=[round((e-s)/1e3) for s,e,n in rows if 'k_fps' in n]
print('GF_OVERLAP=$ov bfs', bfs[8:24]); print('   fps', fps[8:24])
PY
done
