for q in 16 4; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues $q main', d['value'], 'nq128', d['secondary']['nq128_train_yaml_eval_forward']['ms_per_step'], 'train', d['secondary']['train_step_b4']['full_step']['ms_per_step'])
"
done
