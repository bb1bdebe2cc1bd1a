# same-box A/B of the training step: the tree's library against another build of the same ABI.  usage: r6_ab_lib.sh path/libasm_hip.so [reps]
OLD_LIB=$1; REPS=${2:-3}
for rep in $(seq $REPS); do
  for v in new old; do
    if [ $v = old ]; then export ASM_HIP_LIB=$OLD_LIB; else unset ASM_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recipe --no-gradsync 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['step']
print('$v', d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], '3x3', s['conv3x3_class']['ms_per_step'], '1x1', s['conv1x1_class']['ms_per_step'], 'bn', s['bn_class']['ms_per_step'], 'single', d.get('single_stream',{}).get('ms_per_step'))"
  done
done
