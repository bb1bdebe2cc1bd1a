#!/usr/bin/env python
"""Per-kernel HBM traffic of whole training steps from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB;
FETCH_SIZE x 2 for the 16-B/lane access pattern as calibrated in profiles/round1_pmc_traffic.json).
usage: pmc_traffic_step.py FETCH_DIR WRITE_DIR STEPS"""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
  name = re.sub(r'^void ', '', name)
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  m = re.match(r'([A-Za-z_0-9:]+(?:<[^(]*>)?)', name)
  return (m.group(1) if m else name)[:64]


def load(d, counter):
  acc = defaultdict(float); n = defaultdict(int)
  for f in glob.glob(d + '/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
      if row['Counter_Name'] == counter:
        k = short(row['Kernel_Name']); acc[k] += float(row['Counter_Value']); n[k] += 1
  return acc, n


def main():
  fd, wd, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
  f, n = load(fd, 'FETCH_SIZE'); w, _ = load(wd, 'WRITE_SIZE')
  rows = sorted(((2 * f[k] + w.get(k, 0.0), k) for k in f), reverse=True)
  tot = sum(r[0] for r in rows)
  print('| kernel | launches/step | read MB/step | written MB/step | total MB/step |\n|---|---:|---:|---:|---:|')
  for t, k in rows[:45]:
    print('| `%s` | %.0f | %.0f | %.0f | %.0f |' % (k, n[k] / steps, 2 * f[k] / 1024 / steps, w.get(k, 0) / 1024 / steps, t / 1024 / steps))
  print('| **all kernels** | | %.0f | %.0f | **%.0f** |' % (2 * sum(f.values()) / 1024 / steps, sum(w.values()) / 1024 / steps, tot / 1024 / steps))


if __name__ == '__main__':
  main()
