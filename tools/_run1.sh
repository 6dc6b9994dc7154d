cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x16
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet_exec.py tests/test_gpu_spconv.py -x -q 2>&1 | tail -2
for c in 1 2; do
timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/x16/b.log 2>&1
python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/x16/b.log') if l.startswith('{')][-1])
r=d['roofline']; print(d['value'], r['frac'], r['us_per_launch'], d['roofline_convs']['by_level']['1']['us'])
PY
done
