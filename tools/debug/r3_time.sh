# is the slow eager run host-bound?  wall / user / sys seconds and voluntary / involuntary context switches of round 3's bench
mkdir -p gpurun_out
out=$PWD/gpurun_out/r3_time.log
: > $out
for i in 1 2 3 4 5; do
  ( cd _r3 && /usr/bin/time -f "wall %e user %U sys %S vcs %w ics %c" -o /tmp/t.$i python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-gradsync 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out; cat /tmp/t.$i >> $out )
done
cat $out
