bash tools/debug/r6_final.sh round6_d
bash tools/debug/r6_ab_r5.sh > gpurun_out/round6_d_ab_r5.log 2>&1; cat gpurun_out/round6_d_ab_r5.log
