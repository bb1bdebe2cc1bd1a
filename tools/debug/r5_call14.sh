#!/bin/bash
O=gpurun_out/r5_14
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/summary.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -12
grep -A14 "slowest" $O/gpu_tests.log | cut -c1-140
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -1 $O/bench.log | cut -c1-200
cat $O/summary.txt
