#!/bin/bash
# round 5, call 1: sanity of the refactor + sweeps of the ring kernels + a same-box bench line
set -x
O=gpurun_out/r5_1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q > $O/test_conv.log 2>&1; echo "test_conv rc=$?" >> $O/summary.txt
timeout 900 python tools/gemm1_sweep.py --out $O/gemm1_sweep.json > $O/gemm1_sweep.log 2>&1; echo "gemm1_sweep rc=$?" >> $O/summary.txt
timeout 600 python tools/gemm1_sweep.py --wgrad --codes 2,3,4 --out $O/wgrad_sweep.json > $O/wgrad_sweep.log 2>&1; echo "wgrad_sweep rc=$?" >> $O/summary.txt
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/summary.txt
tail -3 $O/test_conv.log; tail -2 $O/gemm1_sweep.log; tail -2 $O/wgrad_sweep.log; tail -1 $O/bench.log | cut -c1-600
cat $O/summary.txt
