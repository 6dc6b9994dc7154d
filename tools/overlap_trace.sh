# Dev tool: kernel trace of the bench to see whether k_geodesic_bfs_lds overlaps the second k_fps launch
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ov -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ov/*/*kernel_trace.csv")[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:],r['Stream_Id']) for r in csv.DictReader(open(f))]
rows.sort()
last=[i for i,r in enumerate(rows) if 'k_voxelize_fp' in r[2]][-2]
T0=rows[last][0]
for s,e,n,st in rows[last:]:
    if any(k in n for k in ('k_fps','k_geodesic','k_ball','k_group','k_knn','k_decoder_cross')):
        print("%8.1f -> %8.1f us  stream %s  %s" % ((s-T0)/1e3,(e-T0)/1e3,st,n))
    if 'k_mask_head' in n: break
PY
