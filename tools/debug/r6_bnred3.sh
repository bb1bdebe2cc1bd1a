# batch-norm backward sums in the 3x3 input gradients (igemm8 / igemm3, operands fetched 8 passes ahead): tests, then the step
# with them (default) against the 1x1 layers only (ASM_BN_RED=1x1), alternating
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "bnred or bn_backward_sums" 2>&1 | tail -3
bash tools/debug/r6_ab_knob.sh ASM_BN_RED=1x1 3
