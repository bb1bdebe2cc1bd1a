#!/bin/bash
P=/root/repo/tools/probes/bin/libasm_prev.so
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh nogemm1=ASM_GEMM1=0 prevA=ASM_HIP_LIB=$P prevAoff=ASM_HIP_LIB=$P,ASM_GEMM1=0 noi3c=ASM_IGEMM3=1 2>&1 | tail -10 | cut -c1-100
mkdir -p gpurun_out/r5_17; cp gpurun_out/ab_knobs.log gpurun_out/r5_17/
