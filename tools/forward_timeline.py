"""Dev tool: timeline of the last full forward in a rocprofv3 kernel trace of bench.py: kernels over 25 us,
idle gaps over 8 us, per stream.   python tools/forward_timeline.py <kernel_trace.csv> [min_kernel_us] [min_gap_us]"""
import csv, sys
f = sys.argv[1]
kmin = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
gmin = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-44:], r['Stream_Id'])
        for r in csv.DictReader(open(f))]
rows.sort()
vf = [i for i, r in enumerate(rows) if 'k_voxelize_fp' in r[2]]
a, b = vf[-3], vf[-2]
T0 = rows[a][0]
print("forward span us", (rows[b][0] - T0) / 1e3, "kernels", b - a)
prev_end, idle = T0, 0.0
for s, e, n, st in rows[a:b]:
    gap = (s - prev_end) / 1e3
    if gap > 0: idle += gap
    if gap > gmin: print("        -- idle %.1f us before %s" % (gap, n))
    if (e - s) / 1e3 > kmin: print("%8.1f -> %8.1f (%6.1f) s%s %s" % ((s - T0) / 1e3, (e - T0) / 1e3, (e - s) / 1e3, st, n))
    prev_end = max(prev_end, e)
print("total idle us", idle)
