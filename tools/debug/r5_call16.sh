#!/bin/bash
for i in 1 2; do
for w in r50 assemble-r152-kd assemble-r50; do
  for g in -1 0; do
    ASM_GEMM1=$g timeout 300 python bench.py --workload $w --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-gradsync --no-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'gemm1=$g', d['value'], d['ms_per_step'], d['step_detail']['gpu_ms_between_step_ends']['median'])"
  done
done
done
