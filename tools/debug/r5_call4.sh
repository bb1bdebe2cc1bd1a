#!/bin/bash
# in-step A/B: one ring configuration forced on every 1x1 layer; 32-channel steps for the Ci = 64 3x3 layers
STEPS=30 WARM=6 bash tools/debug/ab_knobs.sh g10=ASM_GEMM1=10 g5=ASM_GEMM1=5 g12=ASM_GEMM1=12 g14=ASM_GEMM1=14 k3a=ASM_IGEMM_BK32_3X3=1 k3b=ASM_IGEMM_BK32_3X3=2 2>&1 | tail -16 | cut -c1-110
mkdir -p gpurun_out/r5_4; cp gpurun_out/ab_knobs.log gpurun_out/r5_4/
