# same-box A/B of the stream knobs: bash tools/debug/ab_streams.sh (on the GPU box; ~10 s per line)
mkdir -p gpurun_out
run() { # label, env...
  lab=$1; shift
  r=$(env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-gradsync 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$lab $r" >> gpurun_out/ab_streams.log
}
: > gpurun_out/ab_streams.log
for i in 1 2; do
run default A=1
run one_wgrad_stream ASM_WGRAD_STREAMS=1
run no_bl_bwd ASM_BL_BWD=0
run no_bl ASM_BL_STREAMS=0
run no_wgrad_stream ASM_WGRAD_STREAM=0
run single ASM_WGRAD_STREAM=0 ASM_BL_STREAMS=0
done
cat gpurun_out/ab_streams.log
