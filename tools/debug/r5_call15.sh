#!/bin/bash
O=gpurun_out/r5_15
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dp_rccl.py -q -k "allreduce_bucket" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
grep -E "passed|failed|^E " $O/tests.log | tail -6
timeout 600 python tools/gemm1_sweep.py --workload r50 --out $O/gemm1_r50.json > $O/gemm1_r50.log 2>&1; echo "r50 rc=$?" >> $O/summary.txt
tail -1 $O/gemm1_r50.log
timeout 600 python tools/gemm1_sweep.py --workload assemble-r152-kd --batch 128 --out $O/gemm1_r152.json > $O/gemm1_r152.log 2>&1; echo "r152 rc=$?" >> $O/summary.txt
tail -1 $O/gemm1_r152.log
for w in r50 assemble-r152-kd; do
  for g in -1 0; do
    ASM_GEMM1=$g timeout 300 python bench.py --workload $w --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-gradsync --no-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'gemm1=$g', d['value'], d['ms_per_step'])"
  done
done
cat $O/summary.txt
