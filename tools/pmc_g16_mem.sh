# Dev tool: memory-side counters of the level-1 conv kernels (several passes of a few counters each)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
pipe=${1:-0}
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCC_BUSY_avr" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcm_$pipe/g$i
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcm_$pipe/g$i -- python3 $R/tools/prof_conv_g16.py 6 1 1 $pipe > /dev/null 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/pmcm_$pipe/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_g16' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c,v in sorted(acc[k].items()): print('    %-44s %14.0f' % (c, sum(v)/len(v)))
PY
