# same-box A/B of kernel-selection knobs on the whole step: bash tools/debug/ab_knobs.sh LABEL=ENV=VAL[,ENV=VAL] ...
# (on the GPU box; ~12 s per line, every configuration twice, interleaved)
mkdir -p gpurun_out
out=gpurun_out/ab_knobs.log
: > $out
run() { # label, env...
  lab=$1; shift
  r=$(env "$@" python bench.py --steps ${STEPS:-40} --warmup ${WARM:-8} --no-cpu-baseline --no-roofline --no-gradsync --no-recipe $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('streams_autotune',{}).get('side_streams_ms'), d.get('streams_autotune',{}).get('single_stream_ms'), d.get('streams_autotune',{}).get('chosen'), 'gpu', d.get('step_detail',{}).get('gpu_ms_between_step_ends'), 'host', d.get('step_detail',{}).get('host_enqueue_ms'), d.get('step_detail',{}).get('host'))")
  echo "$lab $r" >> $out
}
for i in 1 2; do
  run default A=1
  for spec in "$@"; do
    lab=${spec%%=*}; envs=${spec#*=}
    run $lab $(echo $envs | tr ',' ' ')
  done
done
cat $out
