# Dev tool: rocprofv3 kernel trace + stats of bench.py (no CPU leg); summary of the top kernels -> gpurun_out/<tag>/
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
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
    pass
conv=[r for r in rows if 'k_conv_g16p<1, false' in r['Name']]
if conv:
    import json
    calls=sum(int(r['Calls']) for r in conv); tot=sum(float(r['TotalDurationNs']) for r in conv)
    json.dump({"us_per_launch": round(tot/calls/1e3,3), "calls": calls,
               "per_kernel": {r['Name'].split('(')[0].replace('void ',''): round(float(r['AverageNs'])/1e3,3) for r in conv},
               "source": "profiles/<round>_bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats of bench.py (tools/bench_trace.sh), "
                         "call-weighted mean of the level-1 16->16 launches of the residual blocks (k_conv_g16p<1, false, *>)"},
              open('$R/gpurun_out/$tag/rocprof_conv_l1.json','w'), indent=1)
fam = ('k_conv_g16p', 'k_conv_lw', 'k_conv_os', 'k_conv_flat', 'k_conv_pair', 'k_concat2_idn')
nf = [int(r['Calls']) for r in rows if 'k_voxelize_fp' in r['Name']]
if nf:
    import json
    ftot = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in fam))
    fcalls = sum(int(r['Calls']) for r in rows if any(k in r['Name'] for k in fam))
    json.dump({"us_per_forward": round(ftot / nf[0] / 1e3, 1), "launches_per_forward": round(fcalls / nf[0], 2), "forwards": nf[0],
               "source": "profiles/<round>_bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats of bench.py (tools/bench_trace.sh), summed "
                         "kernel durations of every k_conv_* launch and k_concat2_idn per forward (forwards = k_voxelize_fp calls)"},
              open('$R/gpurun_out/$tag/rocprof_conv_family.json', 'w'), indent=1)
for r in rows[:int('${2:-28}')]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f} total {float(r['TotalDurationNs'])/1e6:7.3f} ms {float(r['Percentage']):5.1f}%")
PY
