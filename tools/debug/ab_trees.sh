# same-box A/B of two TREES: round 3's export under _r3/ (its own library build) against this tree, alternating processes.
# bash tools/debug/ab_trees.sh (on the GPU box) -> gpurun_out/ab_trees.log: label, images/sec, ms per step
mkdir -p gpurun_out
out=$PWD/gpurun_out/ab_trees.log
: > $out
run() { # label, dir, extra args...
  lab=$1; dir=$2; shift; shift
  r=$(cd $dir && python bench.py --steps ${STEPS:-30} --warmup ${WARM:-5} --no-cpu-baseline --no-roofline --no-gradsync "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('step_detail') or {}).get('host'))")
  echo "$lab $r" >> $out
}
for i in 1 2 3; do
  run round3_tree _r3
  run this_tree_recorded_step .
  run this_tree_eager . --eager
done
cat $out
