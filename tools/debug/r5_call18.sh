#!/bin/bash
P=/root/repo/tools/probes/bin/libasm_tableA.so
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh nogemm1=ASM_GEMM1=0 tableA=ASM_HIP_LIB=$P 2>&1 | tail -6 | cut -c1-100
mkdir -p gpurun_out/r5_18; cp gpurun_out/ab_knobs.log gpurun_out/r5_18/
