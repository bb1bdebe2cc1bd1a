#!/bin/bash
O=gpurun_out/r5_13
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "igemm3" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
grep -E "passed|failed|^E " $O/tests.log | tail -6
timeout 300 python tools/conv_bench.py --only C64-K128-3x3 --kinds fprop,dgrad --iters 30 2>/dev/null | tail -4
ASM_IGEMM3=3 timeout 300 python tools/conv_bench.py --only C64-K128-3x3 --kinds fprop,dgrad --iters 30 2>/dev/null | tail -4
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh i3c64=ASM_IGEMM3=3 2>&1 | tail -5 | cut -c1-100
cp gpurun_out/ab_knobs.log $O/
cat $O/summary.txt
