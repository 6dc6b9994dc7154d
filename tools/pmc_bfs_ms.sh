# Dev tool: hardware counters of the multi-source BFS kernels (one rocprofv3 --pmc pass per counter group)
#   bash tools/pmc_bfs_ms.sh   -> gpurun_out/pmc_bfs_ms/summary.txt
export GPU_MAX_HW_QUEUES=16
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_bfs_ms; rm -rf $O; mkdir -p $O
i=0
for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/prof_bfs_ms.py > $O/g$i.log 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/g*/*/*counter_collection.csv') + glob.glob('$O/g*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n.startswith('k_ms'):
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = open('$O/summary.txt', 'w')
for k in sorted(acc):
    m = {c: (sum(v) / len(v), len(v)) for c, v in acc[k].items()}
    s = k + ": " + ", ".join(f"{c}={v[0]:.4g}" for c, v in sorted(m.items())) + f" (n={max(v[1] for v in m.values())})"
    print(s); out.write(s + "\n")
PY
