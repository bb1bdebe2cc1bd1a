#!/bin/bash
P=/root/repo/tools/probes/bin/libasm_wpx32.so
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh wpx32=ASM_HIP_LIB=$P 2>&1 | tail -5 | cut -c1-100
mkdir -p gpurun_out/r5_12; cp gpurun_out/ab_knobs.log gpurun_out/r5_12/
