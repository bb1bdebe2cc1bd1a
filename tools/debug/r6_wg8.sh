# wgrad8_kernel against the register-staged 256 x 256 kernel it replaced, same box.  The baseline library is a worktree build of
# the commit before wgrad8 (tools/debug/r6_ab_r5.sh shows the worktree recipe) passed as OLD_LIB=path/libasm_hip.so.
# 1. the weight-gradient tests  2. per-shape times (tools/conv_bench.py --kinds wgrad)  3. the training step, alternating
OLD_LIB=${OLD_LIB:-$PWD/tools/debug/libvar/wg8off/libasm_hip.so}
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3
SH="--shape 256,14,14,512,1024,3,3,1 --shape 256,7,7,512,1024,3,3,1 --shape 256,28,28,128,256,3,3,1 --shape 256,14,14,256,512,3,3,1 --shape 256,14,14,1024,1024,1,1,1 --shape 256,7,7,256,512,3,3,1 --shape 256,14,14,128,256,3,3,1 --shape 256,28,28,256,512,1,1,1"
for rep in 1 2; do
  echo "== new"; timeout 300 python tools/conv_bench.py --kinds wgrad --iters 20 $SH 2>&1 | tail -10
  echo "== old"; ASM_HIP_LIB=$OLD_LIB timeout 300 python tools/conv_bench.py --kinds wgrad --iters 20 $SH 2>&1 | tail -10
done
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export ASM_HIP_LIB=$OLD_LIB; else unset ASM_HIP_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recipe --no-gradsync 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['step']
print('$v', d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], '3x3', s['conv3x3_class']['ms_per_step'], '1x1', s['conv1x1_class']['ms_per_step'], 'bn', s['bn_class']['ms_per_step'])"
  done
done
