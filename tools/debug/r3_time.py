"""Is a slow eager run host-bound?  Runs round 3's bench (under _r3/) a few times and prints, per run, its images/sec, ms
per step and the CPU seconds (user, sys) and context switches the process used, from getrusage."""
import json, os, resource, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tree = sys.argv[1] if len(sys.argv) > 1 else '_r3'
extra = sys.argv[2:]
prev = resource.getrusage(resource.RUSAGE_CHILDREN)
for i in range(5):
  t0 = time.time()
  p = subprocess.run([sys.executable, 'bench.py', '--steps', '40', '--warmup', '5', '--no-cpu-baseline', '--no-roofline',
                      '--no-gradsync'] + extra, cwd=os.path.join(root, tree), capture_output=True, text=True)
  wall = time.time() - t0
  cur = resource.getrusage(resource.RUSAGE_CHILDREN)
  try:
    d = json.loads(p.stdout.strip().splitlines()[-1])
    res = '%s %s' % (d['value'], d['ms_per_step'])
  except Exception as e:
    res = 'ERR %r %s' % (e, p.stderr[-300:])
  print('%s: %s | wall %.1f user %.1f sys %.1f vcs %d ics %d' % (tree, res, wall, cur.ru_utime - prev.ru_utime,
        cur.ru_stime - prev.ru_stime, cur.ru_nvcsw - prev.ru_nvcsw, cur.ru_nivcsw - prev.ru_nivcsw), flush=True)
  prev = cur
