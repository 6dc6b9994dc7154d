# Dev tool: kernel trace + per-ms timeline of the calibrated batch-4 training step (tools/prof_train_dp.py)
R=$GRAFT_REPO_ROOT; tag=${1:-r4_tdp}; mkdir -p $R/gpurun_out/$tag; cd $R
timeout 300 python3 tools/prof_train_dp.py 8 2>&1 | grep "^step" > gpurun_out/$tag/step.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/prof_train_dp.py 4 > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(ls gpurun_out/$tag/prof/*/*kernel_trace.csv | head -1)
python3 tools/train_timeline.py $f 1.0 -2 > gpurun_out/$tag/timeline.txt 2>&1
cp $(ls gpurun_out/$tag/prof/*/*kernel_stats.csv | head -1) gpurun_out/$tag/kernel_stats.csv
cp $f gpurun_out/$tag/kernel_trace.csv
rm -rf gpurun_out/$tag/prof
cat gpurun_out/$tag/step.txt
