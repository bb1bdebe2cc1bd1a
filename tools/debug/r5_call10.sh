#!/bin/bash
O=gpurun_out/r5_10
mkdir -p $O
P=/root/repo/tools/probes/bin/libasm_prev.so
STEPS=30 WARM=8 bash tools/debug/ab_knobs.sh prevA=ASM_HIP_LIB=$P prevTies=ASM_HIP_LIB=$P,ASM_GEMM1=-2 2>&1 | tail -9 | cut -c1-100
cp gpurun_out/ab_knobs.log $O/
timeout 600 python tools/insitu_sweep.py --out $O/insitu.json > $O/insitu.log 2>&1
tail -14 $O/insitu.log | cut -c1-160
