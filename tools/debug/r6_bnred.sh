timeout 900 python -m pytest tests/test_gpu_conv.py -q -k "bn_backward_sums or igemm8" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "r152_forward" 2>&1 | tail -5
for v in 1 0 2 1 0; do echo "== ASM_BN_RED=$v"; ASM_BN_RED=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recipe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], json.dumps(d['step'].get('bn_class')), json.dumps(d['step'].get('conv1x1_class'))[:200], json.dumps(d['step'].get('conv3x3_class'))[:160])"; done
