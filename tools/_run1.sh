R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for G in 1 2 4 8; do
rm -rf $R/gpurun_out/lv1; mkdir -p $R/gpurun_out/lv1
GF_CONV_GA_G=$G timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lv1/prof -- python3 $R/tools/prof_conv_levels.py 6 > $R/gpurun_out/lv1/log 2>&1
echo "G=$G"; python3 $R/tools/summarize_trace.py $(ls $R/gpurun_out/lv1/prof/*/*kernel_trace.csv) k_conv_ga | tail -3
done
