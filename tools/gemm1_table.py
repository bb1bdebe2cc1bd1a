#!/usr/bin/env python
"""gemm1_sweep.py JSON(s) -> rows of the per-layer table of csrc/conv_gemm1.hip (kAuto) that are not in it yet: the best
configuration where it beats igemm2 by >= 4 % stand-alone, or -- the footprint argument of DESIGN.md section 5.0 -- ties it
within 3 %.  usage: gemm1_table.py batch sweep.json [batch2 sweep2.json ...]"""
import json
import re
import sys

src = open(__file__.replace('tools/gemm1_table.py', 'assembled_cnn_amd/csrc/conv_gemm1.hip')).read()
have = {tuple(map(int, m.groups()[:4])) for m in re.finditer(r'\{(\d), +(\d+), +(\d+), +(\d+), +(\d+)\}', src)}
kid = {'fprop': 0, 'dgrad': 1, 'dgrad+add': 2}
rows = []
args = sys.argv[1:]
for batch, path in zip(args[0::2], args[1::2]):
  d = json.load(open(path))
  for r in d['rows']:
    t = r['times']
    base = sum(t['0'])
    alt = sorted((sum(v), c) for c, v in t.items() if c != '0')
    if not alt or alt[0][0] > 1.03 * base:
      continue
    k = kid[r['kind']]
    M = int(batch) * r['H'] * r['H']
    ci, co = (r['C'], r['K']) if k == 0 else (r['K'], r['C'])
    key = (k, M, ci, co)
    if key in have:
      continue
    have.add(key)
    rows.append('{%d, %6d, %4d, %4d, %2d}' % (k, M, ci, co, int(alt[0][1])))
print(len(rows), 'new rows')
for i in range(0, len(rows), 4):
  print('    ' + ', '.join(rows[i:i + 4]) + ',')
