#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_bfs_ms.sh
export GPU_MAX_HW_QUEUES=16
R=$PWD
mkdir -p $R/gpurun_out/bfs_ms
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bfs_ms -o t -- python3 $R/tools/prof_bfs_ms.py > $R/gpurun_out/bfs_ms/log.txt 2>&1
cd $R
find gpurun_out/bfs_ms -name "*kernel_stats*" | head
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/bfs_ms/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows = [r for r in rows if r["Kernel_Name"].startswith(("k_ms", "void k_ms"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call only
starts = [i for i, r in enumerate(rows) if "k_ms_count" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
out = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:24], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
for o in out[:14]: print("%-24s start %8.1f dur %6.2f gap %6.2f" % o)
print("...")
for o in out[100:106]: print("%-24s start %8.1f dur %6.2f gap %6.2f" % o)
print("...")
for o in out[-6:]: print("%-24s start %8.1f dur %6.2f gap %6.2f" % o)
hop = [o for o in out if "hop" in o[0]]
import statistics
print("hops", len(hop), "mean dur %.2f mean gap %.2f" % (statistics.mean(o[2] for o in hop), statistics.mean(o[3] for o in hop)))
print("total %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
PY
