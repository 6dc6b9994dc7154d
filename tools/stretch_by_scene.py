"""Dev tool: per forward of a rocprofv3 kernel trace of bench.py -- the sampling / BFS stretch: first sampling launch
(query picks), second sampling launch (rest), BFS, set abstraction end, decoder start.
   python tools/stretch_by_scene.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
vf = [i for i, r in enumerate(rows) if 'k_voxelize_fp' in r[2]]
print("forward | span | backbone end | fps1 start..end | fps2 end | bfs start..end (dur) | group_mlp end | first cross-attn start | last kernel end")
for a, b in zip(vf[8:-1], vf[9:]):
    T0 = rows[a][0]
    seg = rows[a:b]
    us = lambda t: (t - T0) / 1e3
    fps = [r for r in seg if 'k_fps' in r[2]]
    bfs = [r for r in seg if 'geodesic_bfs' in r[2]]
    gm = [r for r in seg if 'k_group_mlp_max' in r[2]]
    ca = [r for r in seg if 'cross_attn' in r[2]]
    if not fps or not bfs or not ca: continue
    end = max(r[1] for r in seg)
    print(f"{us(rows[b][0]):7.0f} | {us(fps[0][0]):6.0f} | {us(fps[0][0]):6.0f}..{us(fps[0][1]):6.0f} | {us(fps[-1][1]):6.0f} | {us(bfs[0][0]):6.0f}..{us(bfs[0][1]):6.0f} ({(bfs[0][1]-bfs[0][0])/1e3:6.0f}) | {us(gm[0][1]) if gm else -1:6.0f} | {us(ca[0][0]):6.0f} | {us(end):6.0f}")
