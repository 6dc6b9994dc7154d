"""Dev tool: every kernel of the last full forward in a kernel trace between the first launch whose name contains
<from> and the first later one containing <to>.   python tools/region_detail.py <trace.csv> <from> <to>"""
import csv, sys
f, a_name, b_name = sys.argv[1], sys.argv[2], sys.argv[3]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-70:], r['Stream_Id'])
        for r in csv.DictReader(open(f))]
rows.sort()
vf = [i for i, r in enumerate(rows) if 'k_voxelize_fp' in r[2]]
a, b = vf[-3], vf[-2]
start = next(i for i in range(a, b) if a_name in rows[i][2])
T0 = rows[start][0]
prev = T0
for s, e, n, st in rows[start:b]:
    print("%7.1f +%6.1f gap %6.1f s%s %s" % ((s - T0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, st, n))
    prev = max(prev, e)
    if b_name in n and s > T0:
        break
