#!/bin/bash
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh i34=ASM_IGEMM3=4 s2off=ASM_DGRAD_S2=0 both=ASM_IGEMM3=4,ASM_DGRAD_S2=0 2>&1 | tail -8 | cut -c1-100
mkdir -p gpurun_out/r5_23; cp gpurun_out/ab_knobs.log gpurun_out/r5_23/
