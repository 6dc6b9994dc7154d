# Dev tool: hardware counters of the forward's kernels, one rocprofv3 --pmc pass per counter group over
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
# tools/prof_forward.py (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with other trace
# domains than --kernel-trace).  Kernels are grouped by name and, for the conv kernels, by grid size (= U-Net level).
#   bash tools/pmc_forward.sh <tag> [regex of kernel names]   -> gpurun_out/<tag>/summary.md
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
tag=${1:-pmc_fwd}; pat=${2:-"k_conv_os|k_conv_flat|k_conv_g16p|k_decoder_cross_attn|k_mask_head|k_geodesic_bfs_lds|k_fps"}
O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/prof_forward.py 3 > $O/g$i.log 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections, re
pat = re.compile(r"$pat")
def key(r):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    g = r.get('Grid_Size') or r.get('Grid_Size_X') or ''
    w = r.get('Workgroup_Size') or r.get('Workgroup_Size_X') or ''
    return f"{n} grid={g} wg={w}" if 'k_conv' in n else n
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('$O/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat.search(r['Kernel_Name']):
            acc[key(r)][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/g1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if pat.search(r['Kernel_Name']):
            dur[key(r)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = ["| kernel | launches | us (profiled) | FETCH MB (x2) | WRITE MB | L2 hit % | MFMA busy % | wait-any % | VALU / SALU / MFMA / LDS / VMEM insts per wave | waves |", "|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    g = lambda c: m.get(c, float('nan'))
    d = dur.get(k, [])
    hit = 100 * g('TCC_HIT_sum') / max(1.0, g('TCC_HIT_sum') + g('TCC_MISS_sum'))
    wv = max(1.0, g('SQ_WAVES'))
    lines.append(f"| \`{k}\` | {len(d) // 1} | {sum(d) / max(1, len(d)):.1f} | {2 * g('FETCH_SIZE') * 1024 / 1e6:.1f} | {g('WRITE_SIZE') * 1024 / 1e6:.1f} | {hit:.0f} | "
                 f"{100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1.0, g('SQ_BUSY_CYCLES')):.0f} | {100 * g('SQ_WAIT_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):.0f} | "
                 f"{g('SQ_INSTS_VALU') / wv:.0f} / {g('SQ_INSTS_SALU') / wv:.0f} / {g('SQ_INSTS_MFMA') / wv:.0f} / {g('SQ_INSTS_LDS') / wv:.0f} / {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / wv:.0f} | {g('SQ_WAVES'):.0f} |")
raw = ["", "raw counter means per launch:", ""]
for k in sorted(acc):
    raw.append(f"* \`{k}\`: " + ", ".join(f"{c}={v:.0f}" for c, v in sorted({c: sum(v) / len(v) for c, v in acc[k].items()}.items())))
open('$O/summary.md', 'w').write("\n".join(lines + raw) + "\n")
print("\n".join(lines))
PY
rm -rf $O/g*/
