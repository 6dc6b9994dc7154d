# Dev tool (round 4): the driver's two commands on the current tree -> gpurun_out/<tag>/
#   pytest -m gpu -x -q ; python3 bench.py --gpus 1 --steps 20 --warmup 5  (+ the staggered loop under the same arguments)
tag=${1:-r4}; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/$tag/gputest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/$tag/gputest.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc $?"
tail -3 gpurun_out/$tag/bench.err; cut -c1-160 gpurun_out/$tag/bench.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --staggered --no-cpu-baseline --no-secondary > gpurun_out/$tag/bench_staggered.json 2>> gpurun_out/$tag/bench.err
cut -c1-160 gpurun_out/$tag/bench_staggered.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/$tag/bench.json'))
print('headline', d['value'], d['ms_per_step'])
for k,v in d.get('secondary',{}).items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','ms_per_episode','ms_per_requery','full_step')})
for k in ('roofline','roofline_convs','roofline_decoder','roofline_mask_head','roofline_bfs','sampling'):
    v=d.get(k) or {}
    print(k, {a:v.get(a) for a in ('frac','us_per_launch','us_per_forward','us_per_pick')})
print('parity', d.get('parity_s150k'))
PY
