#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, per (short) kernel name.
usage: pmc_summary.py DIR [DIR ...] [--match SUBSTR] [--md TITLE]   (--md: the markdown table committed under profiles/)"""
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
  md = None
  if '--md' in sys.argv:
    md = sys.argv[sys.argv.index('--md') + 1]
    args = [a for a in args if a != md]
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
  if md is not None:
    def mean(k, c):
      v = acc[k].get(c, [0.0])
      return sum(v) / len(v)
    print('# %s: SQ counters per kernel (one rocprofv3 --pmc pass, no trace domains)\n' % md)
    print('`rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA '
          'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`; means per dispatch.')
    print('MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): the fraction of the kernel\'s duration '
          'an average SIMD\'s matrix pipe was executing (32 cycles per v_mfma_f32_32x32x16_bf16).')
    print('issuing / issue-stalled / parked = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES (disjoint shares of a '
          'wave\'s lifetime: issuing, stalled on a dependency or pipe, parked on s_waitcnt / barrier).\n')
    print('| kernel | dispatches | avg us (under PMC) | MFMA busy | issuing | issue-stalled | parked | waves / dispatch |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    order = sorted(acc, key=lambda k: -sum(dur[k]))
    for k in order:
      if k.startswith('at::') or k.startswith('__amd'):
        continue
      wc = mean(k, 'SQ_WAVE_CYCLES') or 1.0
      gui = mean(k, 'GRBM_GUI_ACTIVE') or 1.0
      busy = mean(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / (gui / 8.0)
      print('| `%s` | %d | %.1f | %.1f %% | %.0f %% | %.0f %% | %.0f %% | %.0f |' % (
          k, len(dur[k]), sum(dur[k]) / len(dur[k]), 100 * busy, 100 * mean(k, 'SQ_ACTIVE_INST_ANY') / wc,
          100 * mean(k, 'SQ_WAIT_INST_ANY') / wc, 100 * mean(k, 'SQ_WAIT_ANY') / wc, mean(k, 'SQ_WAVES')))
    return
  for k in sorted(acc):
    print('== %s  (%d dispatches, avg %.1f us under PMC)' % (k, len(dur[k]), sum(dur[k]) / max(1, len(dur[k]))))
    for c in sorted(acc[k]):
      v = acc[k][c]
      print('   %-28s %16.1f' % (c, sum(v) / len(v)))


if __name__ == '__main__':
  main()
