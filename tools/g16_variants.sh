# Dev tool: kernel-trace tools/prof_conv_g16.py for the base library and every variant under geoformer_amd/lib/variants
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in base $(ls $R/geoformer_amd/lib/variants/*.so 2>/dev/null); do
  n=$(basename $v .so)
  if [ "$v" != base ]; then export GF_LIB_PATH=$v; fi
  for cfg in "1 1" "0 1" "1 2"; do
  rm -rf $R/gpurun_out/g16_$n
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g16_$n -- python3 $R/tools/prof_conv_g16.py 20 $cfg > /dev/null 2>&1
  echo "== $n ldsw/gpw $cfg: $(python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/g16_$n/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_conv_g16' in r['Name']:
        print(r['Name'].split('(')[0][5:40], r['Calls'], round(float(r['AverageNs'])/1e3,2), 'min', round(float(r['MinNs'])/1e3,2), end=' | ')
PY
)"
  done
done
