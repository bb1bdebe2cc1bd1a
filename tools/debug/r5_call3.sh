#!/bin/bash
# A/B of the round-5 kernel selections inside the recorded step (same box, interleaved, twice)
bash tools/debug/ab_knobs.sh nogemm1=ASM_GEMM1=0 noring=ASM_WGRAD_RING=0 neither=ASM_GEMM1=0,ASM_WGRAD_RING=0 2>&1 | tail -12
mkdir -p gpurun_out/r5_3; cp gpurun_out/ab_knobs.log gpurun_out/r5_3/
