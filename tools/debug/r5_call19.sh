#!/bin/bash
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh ws1=ASM_WGRAD_STREAMS=1 ws3=ASM_WGRAD_STREAMS=3 sc=ASM_SC_STREAM=1 nobl=ASM_BL_BWD=0 2>&1 | tail -10 | cut -c1-100
mkdir -p gpurun_out/r5_19; cp gpurun_out/ab_knobs.log gpurun_out/r5_19/
