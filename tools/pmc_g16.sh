# Dev tool: SQ / GRBM counters of the level-1 conv kernels (one pass), pipelined and not
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for pipe in 0 1; do
rm -rf $R/gpurun_out/pmc_g16_$pipe
timeout 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_g16_$pipe -- python3 $R/tools/prof_conv_g16.py 10 1 1 $pipe > /dev/null 2>&1
echo "pipe=$pipe rc=$?"
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_g16_$pipe/*/*counter_collection.csv')
kt=glob.glob('$R/gpurun_out/pmc_g16_$pipe/*/*kernel_trace.csv')
dur=collections.defaultdict(list)
for r in csv.DictReader(open(kt[0])):
    if 'k_conv_g16' in r['Kernel_Name']: dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if 'k_conv_g16' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    d=sum(dur[k])/len(dur[k])
    print(k, 'avg us', round(d,2))
    for c,v in acc[k].items(): print('   ',c, round(sum(v)/len(v)))
    g=sum(acc[k]['GRBM_GUI_ACTIVE'])/len(acc[k]['GRBM_GUI_ACTIVE'])
    print('    eff clock GHz', round(g/8/d/1e3,3), ' mfma busy/SIMD-cycles', round(sum(acc[k]['SQ_VALU_MFMA_BUSY_CYCLES'])/len(acc[k]['SQ_VALU_MFMA_BUSY_CYCLES'])/1024))
PY
done
