# resident-row weight gradient as the default: every conv shape of the workload at batch 256 against the direct kernels, the
# conv -> BN groups at batch 256 that use it, then the whole step with the knob off / on (bash, on the GPU box)
mkdir -p gpurun_out
out=gpurun_out/t_rows2.log
timeout 300 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -m gpu -k "batch_256 and assemble" 2>&1 | grep -E "passed|failed|rror|wgrad" | head -8 > $out
timeout 200 python -m pytest tests/test_gpu_groups_n256.py -x -q -m gpu -k "14x14x512 or 7x7x512 or 14x14x256" 2>&1 | grep -E "passed|failed|rror" | head -4 >> $out
STEPS=20 WARM=5 bash tools/debug/ab_knobs.sh rows_off=ASM_WGRAD_ROWS=0 > /dev/null 2>&1
cut -c1-60 gpurun_out/ab_knobs.log >> $out
cat $out
