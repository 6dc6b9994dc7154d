# Dev tool: HBM-side traffic and issue counters of the conv kernels whose rooflines are quoted (k_conv_g16p: level-1
# 16 -> 16; k_conv_lw: level-2 32 -> 32 and level-1 32 -> 16), one rocprofv3 --pmc pass per counter group (FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Writes gpurun_out/pmc_conv/summary.md and pmc_conv_l1_latest.json (copy both into profiles/).
export GPU_MAX_HW_QUEUES=16  # (in this shell: the profiler brings the GPU up before python starts)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/prof_conv_kernels.py 6 > /dev/null 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, hashlib
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('$O/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_g16p' in r['Kernel_Name'] or 'k_conv_lw' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/g1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_g16p' in r['Kernel_Name'] or 'k_conv_lw' in r['Kernel_Name']:
            dur[r['Kernel_Name'].split('(')[0].replace('void ', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = ["| kernel | counter | per launch |", "|---|---|---|"]
out = {}
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    lines.append(f"| \`{k}\` | duration under the profiler (us) | {sum(dur[k]) / max(1, len(dur[k])):.2f} |")
    for c in sorted(m):
        lines.append(f"| \`{k}\` | {c} | {m[c]:.0f} |")
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        raw = (m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
        cor = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
        lines.append(f"| \`{k}\` | HBM-side bytes: raw FETCH+WRITE (KiB units) / with the gfx950 x2 on FETCH_SIZE | {raw / 1e6:.1f} MB / {cor / 1e6:.1f} MB |")
        out[k] = {"fetch_kib": m['FETCH_SIZE'], "write_kib": m['WRITE_SIZE'], "bytes_raw": int(raw), "bytes_corrected": int(cor)}
open('$O/summary.md', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
if out:
    tot = [v["bytes_corrected"] for kk, v in out.items() if "k_conv_g16p" in kk]
    json.dump({"bytes": int(sum(tot) / len(tot)), "per_kernel": out,
               "kernel_source_sha256": hashlib.sha256(open('$R/geoformer_amd/csrc/spconv_conv.hip', 'rb').read()).hexdigest(),
               "source": "profiles/r6_pmc_conv.md (tools/pmc_conv.sh: separate --pmc passes, FETCH_SIZE x2 per MI355X_MICROARCH.md, mean of the two launches a forward issues: residual epilogue and activation epilogue)"},
              open('$O/pmc_conv_l1_latest.json', 'w'), indent=1)
PY
