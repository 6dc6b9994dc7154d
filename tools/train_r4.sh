# Dev tool: the batch-4 training step at the end of round 4: stage times, host-bound check, launch counts, kernel trace + timeline
export GPU_MAX_HW_QUEUES=16  # (in this shell: under rocprofv3 the profiler brings the GPU up before python starts)
R=$GRAFT_REPO_ROOT; tag=${1:-r4_trn}; mkdir -p $R/gpurun_out/$tag; cd $R
PHASES=1 HOSTBOUND=1 OPCOUNT=1 timeout 400 python3 tools/prof_train_step.py 6 > gpurun_out/$tag/phases.txt 2>&1
BWDNAMES=bwd timeout 300 python3 tools/prof_train_step.py 2 > gpurun_out/$tag/bwdnames.txt 2>&1
BWDNAMES=fwd timeout 300 python3 tools/prof_train_step.py 2 > gpurun_out/$tag/fwdnames.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/prof_train_step.py 4 > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
f=$(ls gpurun_out/$tag/prof/*/*kernel_trace.csv | head -1)
python3 tools/train_timeline.py $f 1.0 -2 > gpurun_out/$tag/timeline.txt 2>&1
cp $(ls gpurun_out/$tag/prof/*/*kernel_stats.csv | head -1) gpurun_out/$tag/kernel_stats.csv
cp $f gpurun_out/$tag/kernel_trace.csv
rm -rf gpurun_out/$tag/prof
ls -la gpurun_out/$tag
