# Dev tool: rocprofv3 kernel trace + stats of bench.py (no CPU leg); summary of the top kernels -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; tag=${1:-bt}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/$tag; mkdir -p $R/gpurun_out/$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-secondary > $R/gpurun_out/$tag/bench.log 2>&1
tail -1 $R/gpurun_out/$tag/bench.log | cut -c1-400
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/$tag/prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms', round(tot/1e6,2))
for r in rows[:int('${2:-28}')]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f} total {float(r['TotalDurationNs'])/1e6:7.3f} ms {float(r['Percentage']):5.1f}%")
PY
