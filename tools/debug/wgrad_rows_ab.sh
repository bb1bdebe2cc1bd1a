# resident-row weight gradient (asm_tuning.wgrad_rows) against the general kernel: parity test, then per-layer times of the
# 3x3 layers on the 14- and 7-wide maps with the knob off / on.  bash tools/debug/wgrad_rows_ab.sh (on the GPU box)
mkdir -p gpurun_out
out=gpurun_out/t_rows.log
timeout 200 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "resident_row" 2>&1 | grep -E "passed|failed|rror|assert" | head -8 > $out
for k in 0 1; do
  for sel in 3x3-H14 3x3-H7; do
    echo "== ASM_WGRAD_ROWS=$k $sel" >> $out
    ASM_WGRAD_ROWS=$k timeout 100 python tools/conv_bench.py --only $sel --kinds wgrad --iters 30 2>&1 | grep "^x\|weighted" >> $out
  done
done
cat $out
