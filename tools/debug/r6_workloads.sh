# every BASELINE workload of bench.py once, short: value, ms per step, kernels per step, final cross entropy, step mode
for w in r50 assemble-r50-mixup assemble-r50-recipe assemble-r152-kd; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-recipe --no-gradsync 2>gpurun_out/wl_$w.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], d['config']['final_cross_entropy'], d.get('step_mode'))" || tail -3 gpurun_out/wl_$w.err
done
