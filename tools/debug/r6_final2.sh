bash tools/debug/r6_final.sh ${1:-round6_e}
bash tools/debug/r6_ab_r5.sh > gpurun_out/${1:-round6_e}_ab_r5.log 2>&1; cat gpurun_out/${1:-round6_e}_ab_r5.log
