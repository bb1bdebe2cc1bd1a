#!/bin/bash
O=gpurun_out/r5_21
mkdir -p $O
timeout 600 python bench.py --workload assemble-r50-recipe --steps 20 --warmup 8 --no-cpu-baseline > $O/bench_recipe.log 2>$O/bench_recipe.err; echo "recipe bench rc=$?" >> $O/summary.txt
tail -1 $O/bench_recipe.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['step_mode'][:80]); print('eager', d.get('eager_step',{}).get('ms_per_step'), 'dp', d.get('dp',{}).get('ms_per_step_with_exchange'), d.get('dp',{}).get('step_mode'), 'single', d.get('single_stream',{}).get('ms_per_step'))
print(d['config'])"
tail -3 $O/bench_recipe.err
cat $O/summary.txt
