"""Dev tool: kernels of the last repetition in a rocprofv3 kernel trace, delimited by a marker kernel name
(default k_voxelize_fp): start, duration, stream, gaps.   python tools/trace_span.py <kernel_trace.csv> [marker] [min_us]"""
import csv, sys
f = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_voxelize_fp"
kmin = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # which repetition, counted from the end
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-48:], r['Stream_Id'])
        for r in csv.DictReader(open(f))]
rows.sort()
vf = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = vf[-1 - back], vf[-back]
T0 = rows[a][0]
pe = T0
busy = 0
for s, e, n, st in rows[a:b]:
    gap = (s - pe) / 1e3
    if (e - s) / 1e3 >= kmin or gap > 5:
        print("%8.1f %7.1f %s s%s %s" % ((s - T0) / 1e3, (e - s) / 1e3, ("gap %5.1f" % gap) if gap > 3 else "         ", st, n))
    if e > pe:
        busy += e - max(s, pe); pe = e
print("span %.1f us, busy %.1f us, kernels %d" % ((pe - T0) / 1e3, busy / 1e3, b - a))
