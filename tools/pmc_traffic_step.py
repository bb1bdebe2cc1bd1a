#!/usr/bin/env python
"""Per-kernel HBM traffic of whole training steps from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB;
FETCH_SIZE x 2 for the 16-B/lane access pattern as calibrated in profiles/round1_pmc_traffic.json).
usage: pmc_traffic_step.py FETCH_DIR WRITE_DIR STEPS [CLASS_JSON]
With CLASS_JSON the per-class totals bench.py's roofline object quotes are written too: the 3x3-convolution class and
the batch-norm family, in HBM bytes per step."""
import csv, glob, json, re, sys
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


def conv3x3(k):
  """3x3 convolution class by kernel name: igemm2 with a multi-tap filter that is not a stem row filter (3x3 and the
  1x2 / 2x1 / 2x2 parity sub-filters of a stride-2 3x3 input gradient), the resident-halo kernels, and the general
  (non linear-address) weight-gradient kernels (wgrad_kernel, wgrad8_kernel), which on this network run the 3x3 layers and the two stems."""
  m = re.match(r'igemm2?_kernel<([^>]*)', k)
  if m:
    a = [t.strip() for t in m.group(1).split(',')]
    if len(a) >= 9 and a[7].isdigit() and a[8].isdigit():
      r, q = int(a[7]), int(a[8])
      return r * q > 1 and q != 1 or (r, q) == (2, 1)
    return False
  return (k.startswith('conv_halo_kernel') or k.startswith('wgrad_halo_kernel') or k.startswith('igemm3_kernel') or
          bool(re.match(r'wgrad_kernel<\d+, \d+, false', k)) or k.startswith('wgrad8_kernel<false'))


def bn_family(k):
  return k.startswith(('bn_', 'rowreduce', 'partials_compact', 'sk_bn_bwd'))


def main():
  fd, wd, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
  f, n = load(fd, 'FETCH_SIZE'); w, _ = load(wd, 'WRITE_SIZE')
  rows = sorted(((2 * f[k] + w.get(k, 0.0), k) for k in f), reverse=True)
  tot = sum(r[0] for r in rows)
  print('| kernel | launches/step | read MB/step | written MB/step | total MB/step |\n|---|---:|---:|---:|---:|')
  for t, k in rows[:45]:
    print('| `%s` | %.0f | %.0f | %.0f | %.0f |' % (k, n[k] / steps, 2 * f[k] / 1024 / steps, w.get(k, 0) / 1024 / steps, t / 1024 / steps))
  print('| **all kernels** | | %.0f | %.0f | **%.0f** |' % (2 * sum(f.values()) / 1024 / steps, sum(w.values()) / 1024 / steps, tot / 1024 / steps))
  if len(sys.argv) > 4:
    byts = {k: (2 * f[k] + w.get(k, 0.0)) * 1024 / steps for k in f}
    out = {'_about': 'HBM bytes per training step by kernel class, from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                     'passes of `bench.py --steps 2 --warmup 1` (tools/profile_round.sh traffic); KiB counters, fetch x 2 '
                     '(MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read on gfx950)',
           'conv3x3_class_bytes_per_step': int(sum(v for k, v in byts.items() if conv3x3(k))),
           'conv3x3_class_kernels': sorted(k for k in byts if conv3x3(k)),
           'bn_class_bytes_per_step': int(sum(v for k, v in byts.items() if bn_family(k))),
           'bn_class_kernels': sorted(k for k in byts if bn_family(k)),
           'all_kernels_bytes_per_step': int(sum(byts.values())),
           'launches_per_step': round(sum(n.values()) / steps, 1)}
    json.dump(out, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
  main()
