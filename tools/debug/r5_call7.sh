#!/bin/bash
O=gpurun_out/r5_7
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_groups_n256.py -k sk_blocks tests/test_gpu_model.py::test_recorded_step_with_kd_and_mixup_type_2 -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
grep -E "passed|failed|^forward|^dx " $O/tests.log | tail -8
# in-step A/B of footprint-related selections (recorded step, interleaved, twice)
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh i3all=ASM_IGEMM3=2 t128=ASM_IGEMM_TILE=1 wsmall=ASM_WGRAD_BIG=0 bn512=ASM_BN_ROWS=512 bn2048=ASM_BN_ROWS=2048 nopfa=ASM_IGEMM_PFA=0 2>&1 | tail -14 | cut -c1-100
cp gpurun_out/ab_knobs.log $O/
cat $O/summary.txt
