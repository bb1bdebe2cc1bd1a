# same-box A/B of stream / runtime knobs: bash tools/debug/ab_streams.sh (on the GPU box)
mkdir -p gpurun_out
run() { # label, env...
  lab=$1; shift
  r=$(env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-gradsync 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$lab $r" >> gpurun_out/ab_streams.log
}
: > gpurun_out/ab_streams.log
for i in 1 2; do
run turn3 ASM_BL_TURN=3
run turn1 ASM_BL_TURN=1
run turn6 ASM_BL_TURN=6
run turn100 ASM_BL_TURN=100
done
cat gpurun_out/ab_streams.log
