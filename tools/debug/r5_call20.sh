#!/bin/bash
O=gpurun_out/r5_20
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/summary.txt
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -8
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
tail -1 $O/smoke.log | cut -c1-300
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -1 $O/bench.log | cut -c1-200
cat $O/summary.txt
