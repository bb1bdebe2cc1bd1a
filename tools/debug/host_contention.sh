# What a busy host does to the step: bench.py pinned to one core, alone and beside 1 / 2 busy loops pinned to the same core.
# (bash tools/debug/host_contention.sh on the GPU box; prints value, ms per step, calibration, per-step GPU / host times)
mkdir -p gpurun_out
out=gpurun_out/host_contention.log
: > $out
run() { # label, competitors
  pids=""
  for i in $(seq 1 $2); do taskset -c 2 python -c "while True: pass" & pids="$pids $!"; done
  r=$(taskset -c 2 python bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-roofline --no-gradsync $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('streams_autotune'), d.get('step_detail'))")
  for p in $pids; do kill $p; done
  echo "$1 $r" >> $out
}
run tape_alone 0
run eager_alone 0 --eager
run tape_two_competitors 2
run eager_two_competitors 2 --eager
run tape_four_competitors 4
cat $out
