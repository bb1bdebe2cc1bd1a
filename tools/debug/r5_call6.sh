#!/bin/bash
O=gpurun_out/r5_6
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/summary.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -30
cat $O/summary.txt
