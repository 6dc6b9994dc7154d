# Dev tool: config-3 training step (batch 4): stage times + rocprofv3 kernel stats of two steps
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/train_profile.py 2>&1 | head -12
rm -rf $R/gpurun_out/train_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_trace -- python3 $R/tools/train_profile.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/train_trace/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms (4 steps)', round(tot/1e6,1))
for r in rows[:32]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
