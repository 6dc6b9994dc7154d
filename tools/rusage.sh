#!/bin/bash
# Dev tool: per-kernel register / LDS / occupancy summary of one HIP source.  tools/rusage.sh spconv_conv [pattern]
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -I$root/include -I$root/geoformer_amd/csrc "${@:3}" \
  -c $root/geoformer_amd/csrc/$1.hip -o /tmp/rusage_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'remark: +(.*?) \[-Rpass',l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur={'name':subprocess.run(['c++filt',t.split(': ')[1]],capture_output=True,text=True).stdout.strip().split('(')[0]}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    if pat in r['name']:
        print(f\"{r['name'][:60]:60s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPR {r.get('TotalSGPRs','?'):>3s} spill {r.get('VGPRs Spill','?'):>3s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?')}\")
" "$2"
