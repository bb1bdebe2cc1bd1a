#!/bin/bash
set -x
O=gpurun_out/r5_2
mkdir -p $O
timeout 900 python tools/gemm1_sweep.py --out $O/gemm1_sweep.json > $O/gemm1_sweep.log 2>&1; echo "gemm1_sweep rc=$?" >> $O/summary.txt
tail -2 $O/gemm1_sweep.log
cat $O/summary.txt
