# same-box A/B of one environment setting against the default, default bench command, alternating.  usage: r6_ab_knob.sh NAME=VALUE [reps]
KV=$1; REPS=${2:-3}
for rep in $(seq $REPS); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $KV; tag="$KV"; else unset ${KV%%=*}; tag="default"; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recipe --no-gradsync 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['step']
print('$tag', d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], '3x3', s['conv3x3_class']['ms_per_step'], '1x1', s['conv1x1_class']['ms_per_step'], 'bn', s['bn_class']['ms_per_step'], 'single', d.get('single_stream',{}).get('ms_per_step'))"
  done
done
