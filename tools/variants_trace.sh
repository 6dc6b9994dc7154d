# Dev tool: kernel-trace a script under every library variant.  usage: variants_trace.sh <script.py> <kernel substring>
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in base $(ls $R/geoformer_amd/lib/variants/*.so 2>/dev/null); do
  n=$(basename $v .so)
  if [ "$v" != base ]; then export GF_LIB_PATH=$v; fi
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vt_$n -- python3 $R/$1 6 > /dev/null 2>&1
  echo "== $n"; python3 $R/tools/summarize_trace.py $(ls $R/gpurun_out/vt_$n/*/*kernel_trace.csv | head -1) $2 | tail -n +3 | awk -F'|' '{print $3, $6}' | tr '\n' ';'; echo
done
