#!/bin/bash
O=gpurun_out/r5_8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -k "kd_and_mixup or records_itself" -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
grep -E "passed|failed|Error|assert " $O/tests.log | tail -8
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh ties=ASM_GEMM1=-2 2>&1 | tail -6 | cut -c1-100
cp gpurun_out/ab_knobs.log $O/
cat $O/summary.txt
