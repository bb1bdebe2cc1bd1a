#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, per (short) kernel name.
usage: pmc_summary.py DIR [DIR ...] [--match SUBSTR]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
  name = re.sub(r'^void ', '', name)
  m = re.match(r'(?:\(anonymous namespace\)::)?([A-Za-z_0-9:]+(?:<[^(]*>)?)', name)
  return (m.group(1) if m else name)[:70]


def main():
  args = [a for a in sys.argv[1:] if not a.startswith('--')]
  match = ''
  if '--match' in sys.argv:
    match = sys.argv[sys.argv.index('--match') + 1]
    args = [a for a in args if a != match]
  acc = defaultdict(lambda: defaultdict(list))
  dur = defaultdict(list)
  for d in args:
    for f in glob.glob(d + '/*counter_collection.csv'):
      seen = set()
      for row in csv.DictReader(open(f)):
        k = short(row['Kernel_Name'])
        if match and match not in k:
          continue
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        key = (f, row['Dispatch_Id'])
        if key not in seen:
          seen.add(key)
          dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
  for k in sorted(acc):
    print('== %s  (%d dispatches, avg %.1f us under PMC)' % (k, len(dur[k]), sum(dur[k]) / max(1, len(dur[k]))))
    for c in sorted(acc[k]):
      v = acc[k][c]
      print('   %-28s %16.1f' % (c, sum(v) / len(v)))


if __name__ == '__main__':
  main()
