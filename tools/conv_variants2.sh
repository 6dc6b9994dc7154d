# Dev tool: level-wise conv timing (BN prologue + residual) for every library variant, all conv kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in base $(ls $R/geoformer_amd/lib/variants/*.so 2>/dev/null); do
  n=$(basename $v .so)
  if [ "$v" != base ]; then export GF_LIB_PATH=$v; fi
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cv2_$n -- python3 $R/tools/prof_conv_levels.py 6 > /dev/null 2>&1
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cv3_$n -- python3 $R/tools/prof_conv_l1.py 6 > /dev/null 2>&1
  echo "== $n"; python3 $R/tools/summarize_trace.py $(ls $R/gpurun_out/cv2_$n/*/*kernel_trace.csv | head -1) k_conv | tail -n +3 | awk -F'|' '{print $2, $3, $6}' | tr '\n' ';'; echo
  echo "   plain:"; python3 $R/tools/summarize_trace.py $(ls $R/gpurun_out/cv3_$n/*/*kernel_trace.csv | head -1) k_conv | tail -n +3 | awk -F'|' '{print $2, $3, $6}' | tr '\n' ';'; echo
done
