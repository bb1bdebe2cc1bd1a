#!/bin/bash
# full GPU suite + the default bench line on the new tree
O=gpurun_out/r5_5
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/summary.txt
tail -15 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -1 $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('step_mode','')[:60])
print('recipe', d.get('recipe'))
print('roofline', {k:v for k,v in d.get('roofline',{}).items() if k in ('frac','achieved')}, d.get('roofline',{}).get('conv1x1'))
print('c11', d.get('step',{}).get('conv1x1_class'))
print('c33', d.get('step',{}).get('conv3x3_class'))
print('bn', d.get('step',{}).get('bn_class'))
print('dp', d.get('dp'))
print('eager', d.get('eager_step',{}).get('ms_per_step'), 'single', d.get('single_stream',{}).get('ms_per_step'), 'launches', d.get('launches',{}).get('kernels_per_step'))
"
tail -5 $O/bench.err
cat $O/summary.txt
