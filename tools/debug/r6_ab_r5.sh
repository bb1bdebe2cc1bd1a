# same-box A/B: the round-5 tree against this tree, default bench command, alternating.  Set-up (once, in the container):
#   git worktree add _r5 7be2d3e && (cd _r5 && python -m assembled_cnn_amd.build)      # _r5/ is git-ignored and travels with gpurun
# then on the GPU box: bash tools/debug/r6_ab_r5.sh
for rep in 1 2 3; do
  for t in _r5 .; do
    (cd $t && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recipe --no-gradsync 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['step']
print('$t', d['value'], d['ms_per_step'], d['launches']['kernels_per_step'], '3x3', s['conv3x3_class']['ms_per_step'], '1x1', s['conv1x1_class']['ms_per_step'], 'bn', s['bn_class']['ms_per_step'], 'single', d.get('single_stream',{}).get('ms_per_step'))")
  done
done
