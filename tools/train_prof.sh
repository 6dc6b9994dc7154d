# Dev tool: the config-3 training step with the fused BatchNorm pair on and off, then a rocprofv3 kernel-stats pass
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
R=$GRAFT_REPO_ROOT; tag=${1:-tp}; mkdir -p $R/gpurun_out/$tag; cd $R
for v in 1 0; do
  GF_FUSED_BN=$v timeout 300 python tools/prof_train_step.py 6 2>&1 | grep "^step" | sed "s/^/GF_FUSED_BN=$v /" | tee -a gpurun_out/$tag/steps.txt
done
GF_FUSED_BN=1 timeout 300 python tools/prof_train_step.py 6 2>&1 | grep "^step" | sed "s/^/GF_FUSED_BN=1 again /" | tee -a gpurun_out/$tag/steps.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/prof_train_step.py 4 > $R/gpurun_out/$tag/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/$tag/prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total GPU ms per step', round(tot/6e6,2), 'launches per step', calls/6)
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} calls/step {int(r['Calls'])/6:7.1f} avg {float(r['AverageNs'])/1e3:8.2f} ms/step {float(r['TotalDurationNs'])/6e6:7.3f}")
PY
