#!/usr/bin/env python
"""profiles/roundN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace
domain) over `tools/conv_bench.py --only C512-K1024-3x3-H14 --kinds fprop,dgrad,wgrad --iters 2` (the layer bench.py's
roofline object names).  Counters are KiB; fetch bytes = 2 x FETCH_SIZE x 1024 for the 16-B/lane access pattern
(MI355X_MICROARCH.md, calibrated in round 1 on the slab reduce whose byte count is known), WRITE_SIZE exact.
usage: pmc_dominant.py FETCH_DIR WRITE_DIR OUT.json"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
  name = re.sub(r'^void ', '', name)
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  m = re.match(r'([A-Za-z_0-9:]+(?:<[^(]*>)?)', name)
  return m.group(1) if m else name


def load(d, counter):
  acc, n = defaultdict(float), defaultdict(int)
  for f in glob.glob(d + '/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
      if row['Counter_Name'] == counter:
        k = short(row['Kernel_Name'])
        acc[k] += float(row['Counter_Value'])
        n[k] += 1
  return {k: acc[k] / n[k] for k in acc}


def main():
  fd, wd, out = sys.argv[1:4]
  f, w = load(fd, 'FETCH_SIZE'), load(wd, 'WRITE_SIZE')
  N, H, C, K = 256, 14, 512, 1024
  alg = {'fprop': 2 * (N * H * H * C + N * H * H * K + 9 * C * K), 'dgrad': 2 * (N * H * H * C + N * H * H * K + 9 * C * K),
         'wgrad': 2 * (N * H * H * C + N * H * H * K) + 4 * 9 * C * K}
  # (round 6: the forward launch is igemm8 for the rows of the full rounds + igemm3 for the last rows: both count)
  groups = {'fprop': [k for k in f if (k.startswith('igemm2_kernel') and ', true, 3, 3' in k) or k.startswith('igemm3_kernel<128, true')
                      or k.startswith('igemm8_kernel<true')],
            'dgrad': [k for k in f if (k.startswith('igemm2_kernel') and ', false, 3, 3' in k) or k.startswith('igemm3_kernel<128, false')
                      or k.startswith('igemm8_kernel<false')],
            'wgrad': [k for k in f if k.startswith('wgrad')]}
  res = {'_about': 'HBM-side traffic per launch, N256 14x14x512 -> 1024 3x3/1, from separate rocprofv3 --pmc FETCH_SIZE / '
                   '--pmc WRITE_SIZE passes over tools/conv_bench.py (see tools/pmc_dominant.py); KiB counters, fetch x 2.'}
  for kind, ks in groups.items():
    if not ks:
      continue
    kern = {k: {'FETCH_SIZE_KiB': round(f.get(k, 0.0), 1), 'WRITE_SIZE_KiB': round(w.get(k, 0.0), 1)} for k in ks}
    tb = sum(2 * v['FETCH_SIZE_KiB'] + v['WRITE_SIZE_KiB'] for v in kern.values()) * 1024
    res['conv %s N256 14x14x512 -> 1024, 3x3/1' % kind] = {'kernels': kern, 'traffic_bytes': int(tb),
                                                          'algorithmic_bytes': alg[kind]}
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
