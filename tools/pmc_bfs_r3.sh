# Dev tool: counters of the geodesic BFS kernel (k_geodesic_bfs_lds) and of the sampler (k_fps), one rocprofv3 --pmc pass
# per counter group over tools/prof_bfs.py (256 sources on a 60k-point foreground).  Writes gpurun_out/pmc_bfs_r3/summary.md
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_bfs_r3; rm -rf $O; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/prof_bfs.py 2 > /dev/null 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('$O/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_geodesic_bfs' in n or 'k_knn_radius' in n:
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/g1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_geodesic_bfs' in n or 'k_knn_radius' in n:
            dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = ["| kernel | counter | per launch |", "|---|---|---|"]
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    lines.append(f"| \`{k}\` | duration under the profiler (us) | {sum(dur[k]) / max(1, len(dur[k])):.1f} |")
    for c in sorted(m):
        lines.append(f"| \`{k}\` | {c} | {m[c]:.0f} |")
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        lines.append(f"| \`{k}\` | HBM-side bytes with the gfx950 x2 on FETCH_SIZE (KiB units) | {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024 / 1e6:.1f} MB |")
open('$O/summary.md', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
