#!/usr/bin/env python
"""gemm1_sweep.py JSON -> a markdown table for profiles/.  usage: sweep_to_md.py in.json 'title' > out.md"""
import json
import sys

d = json.load(open(sys.argv[1]))
title = sys.argv[2] if len(sys.argv) > 2 else 'sweep'
codes = sorted({int(c) for r in d['rows'] for c in r['times']})
print('# %s\n' % title)
print('`tools/gemm1_sweep.py%s` on one MI355X, batch 256, every 1x1 stride-1 shape of Assemble-ResNet-50 + D.  Cells: microseconds per launch,'
      % (' --wgrad' if d.get('field') == 'wgrad_ring' else ''))
print('warm / cold (HIP events; warm = 20 back-to-back launches, cold = each launch behind a 512 MB fill).  Code 0 = the kernel the library')
print('chose before this round (igemm2_kernel / the register-staged weight gradient); the other codes are the configurations of')
print('`%s`.  Outputs are bit-identical to code 0 for every cell shown (fused statistics: same sums in another order).\n'
      % ('wgrad_kernel<.., LIN, NS>: NS = code' if d.get('field') == 'wgrad_ring' else 'csrc/conv_gemm1.hip (asm_gemm1_try)'))
print('| kind | map | C -> K | launches / step | ' + ' | '.join(str(c) for c in codes) + ' | best | vs 0 |')
print('|---|---|---|---:|' + '---:|' * len(codes) + '---:|---:|')
for r in d['rows']:
  t = r['times']
  cells = ['%.1f / %.1f' % tuple(t[str(c)]) if str(c) in t else '-' for c in codes]
  b = str(r['best'])
  print('| %s | %dx%d | %d -> %d | %d | %s | %s | %+.0f %% |' % (r['kind'], r['H'], r['H'], r['C'], r['K'], r['count'], ' | '.join(cells), b,
                                                              100.0 * (sum(t[b]) / sum(t['0']) - 1.0)))
print('\nweighted microseconds per step over these launches (mean of warm and cold): code 0 %.1f, best per shape %.1f (%+.1f %%)'
      % (d['baseline_us'], d['best_us'], 100.0 * (d['best_us'] / d['baseline_us'] - 1.0)))
