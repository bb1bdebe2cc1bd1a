#!/usr/bin/env python
"""How much of a training step does the chip spend nearly empty?  From a rocprofv3 --kernel-trace result (rocpd sqlite,
tools/profile_round.sh stats_default): per replayed step, the time during which NO kernel runs and the time during which only
kernels of fewer than THR workgroups run (a 256-CU chip), with the kernels that own that time.
usage: low_occupancy.py results.db [THR=128] [first_step=5] [steps=3]"""
import collections
import re
import sqlite3
import sys


def short(n):
  n = re.sub(r'\(anonymous namespace\)::', '', n)
  n = re.sub(r'^void ', '', n)
  m = re.match(r'([A-Za-z_0-9:]+(?:<[^(]*>)?)', n)
  return (m.group(1) if m else n)[:60]


def main():
  db = sys.argv[1]
  thr = int(sys.argv[2]) if len(sys.argv) > 2 else 128
  first = int(sys.argv[3]) if len(sys.argv) > 3 else 5
  nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
  c = sqlite3.connect(db)
  rows = c.execute('select name,start,end,grid_x,grid_y,grid_z,workgroup_x,workgroup_y,workgroup_z from kernels order by start').fetchall()
  sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]      # two optimiser launches end a step
  ends = [rows[i][2] for i in sgd[1::2]]

  def wgs(r):
    return (r[3] // max(r[6], 1)) * (r[4] // max(r[7], 1)) * (r[5] // max(r[8], 1))

  print('| step | ms | nothing running | only kernels of < %d workgroups running |' % thr)
  print('|---:|---:|---:|---:|')
  acc, cnt = collections.Counter(), collections.Counter()
  for step in range(first, first + nsteps):
    t0, t1 = ends[step], ends[step + 1]
    ev = []
    for idx, r in enumerate(rows):
      if r[2] <= t0 or r[1] >= t1:
        continue
      big = wgs(r) >= thr
      ev.append((max(r[1], t0), 1, big, idx))
      ev.append((min(r[2], t1), -1, big, idx))
    ev.sort()
    nb, running, last, none, small = 0, set(), t0, 0, 0
    for t, d, big, idx in ev:
      if nb == 0 and running:
        small += t - last
        for r in running:
          acc[short(rows[r][0])] += (t - last) / len(running) / nsteps
      if nb == 0 and not running:
        none += t - last
      last = t
      if big:
        nb += d
      elif d > 0:
        running.add(idx)
        cnt[short(rows[idx][0])] += 1.0 / nsteps
      else:
        running.discard(idx)
    print('| %d | %.3f | %.3f | %.3f |' % (step, (t1 - t0) / 1e6, none / 1e6, small / 1e6))
  print('\n| kernel | ms per step while nothing larger runs | launches per step |\n|---|---:|---:|')
  for k, v in acc.most_common(14):
    print('| `%s` | %.3f | %.0f |' % (k, v / 1e6, cnt[k]))


if __name__ == '__main__':
  main()
