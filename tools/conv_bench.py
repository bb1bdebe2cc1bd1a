#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the workload's own layer shapes (HIP events, GPU only).
usage: conv_bench.py [--top N] [--kinds fprop,dgrad,wgrad] [--workload W] [--batch B] [--iters I]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd import ops  # noqa: E402
from tools.list_convs import conv_shapes  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--top', type=int, default=14)
  ap.add_argument('--kinds', default='fprop,dgrad,wgrad')
  ap.add_argument('--workload', default='assemble-r50')
  ap.add_argument('--batch', type=int, default=256)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--only', default='')
  ap.add_argument('--uniform', action='store_true', help='uniform [-1, 1) operands instead of normal ones')
  ap.add_argument('--shape', action='append', default=[], help='N,H,W,C,K,R,S,stride: bench this shape instead of the workload (repeatable)')
  args = ap.parse_args()
  shapes = [(k, c) for k, c in conv_shapes(args.workload, args.batch).items() if not k[8]]
  if args.shape:
    shapes = [(tuple(int(v) for v in sh.split(',')) + (0,), 1) for sh in args.shape]

  def gf(k):
    N, H, W, Cn, K, R, S, st, _ = k
    Ho = H if st == 1 else (H - 1) // st + 1
    return 2.0 * N * Ho * Ho * K * Cn * R * S / 1e9
  shapes.sort(key=lambda kc: -gf(kc[0]) * kc[1])
  if args.only:
    shapes = [kc for kc in shapes if args.only in 'C%d-K%d-%dx%d-H%d' % (kc[0][3], kc[0][4], kc[0][5], kc[0][6], kc[0][1])]
  shapes = shapes[:args.top]
  g = torch.Generator(device='cuda').manual_seed(0)
  tot = {}
  print('%-34s %5s | %s' % ('shape', 'GF', ' | '.join('%-22s' % k for k in args.kinds.split(','))))
  for k, cnt in shapes:
    N, H, W, Cn, K, R, S, st, _ = k
    d = ops.make_conv_desc(N, H, W, Cn, K, R, S, st)
    rnd = (lambda shp: torch.rand(shp, generator=g, device='cuda') * 2 - 1) if args.uniform else (
        lambda shp: torch.randn(shp, generator=g, device='cuda'))
    x = rnd((N, H, W, Cn)).to(torch.bfloat16)
    w = (rnd((K, R, S, Cn)) * (R * S * Cn) ** -0.5).to(torch.bfloat16)
    dy = rnd((N, d.Ho, d.Wo, K)).to(torch.bfloat16)
    wt = torch.zeros((Cn, R, S, K), dtype=torch.bfloat16, device='cuda')
    ops.filter_transpose(w, wt, K, R, S, Cn)
    dw = torch.empty((K, R, S, Cn), dtype=torch.float32, device='cuda')
    fns = {'fprop': lambda: ops.conv_fprop(d, x, w, True), 'dgrad': lambda: ops.conv_dgrad(d, dy, wt),
           'wgrad': lambda: ops.conv_wgrad(d, x, dy, dw)}
    cells = []
    bytes_io = 2.0 * (x.numel() + dy.numel())
    for kind in args.kinds.split(','):
      fn = fns[kind]
      fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.iters):
        fn()
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.iters
      cells.append('%7.1f us %6.0f TF %4.1f TB/s' % (ms * 1e3, gf(k) / ms, bytes_io / ms / 1e9))
      tot[kind] = tot.get(kind, 0.0) + ms * cnt
    print('%-34s %5.0f | %s' % ('x%d C%d-K%d-%dx%d-H%d/%d' % (cnt, Cn, K, R, S, H, st), gf(k), ' | '.join(cells)))
  print('weighted ms per step over these shapes:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
  main()
