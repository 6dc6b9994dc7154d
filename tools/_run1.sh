cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/x8
timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > gpurun_out/x8/b.log 2>&1; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/x8/b.log') if l.startswith('{')][-1])
print(d['value'], d['secondary'])
PY
