#!/usr/bin/env python
"""Same-box sweep of the weight gradient's pixel-split count (asm_tuning.wgrad_splits) against the cost model's own choice
(csrc/conv_wgrad.hip make_plan), every convolution shape of a workload at the benchmark batch, reduce pass included.
usage: wgrad_split_sweep.py [--workload W] [--batch B] [--splits 2,4,...] [--only SUBSTR] [--out file.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd import lib as _lib, ops  # noqa: E402
from tools.list_convs import conv_shapes  # noqa: E402


def set_field(field, value):
  t = _lib.Tuning()
  ops.L().asm_get_tuning(C.byref(t))
  setattr(t, field, value)
  assert ops.L().asm_set_tuning(C.byref(t)) == 0, ops.L().asm_last_error()


def plan_of(d):
  arr = (C.c_int32 * 6)()
  assert ops.L().asm_conv2d_wgrad_plan(C.byref(d), C.byref(arr)) == 0
  return list(arr)


def time_fn(fn, iters):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='assemble-r50')
  ap.add_argument('--batch', type=int, default=256)
  ap.add_argument('--splits', default='1,2,3,4,6,8,12,16,24,32,48,64,96,128')
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--only', default='')
  ap.add_argument('--out', default='')
  args = ap.parse_args()
  splits = [int(s) for s in args.splits.split(',') if s]
  g = torch.Generator(device='cuda').manual_seed(0)
  rows = []
  for k, cnt in conv_shapes(args.workload, args.batch).items():
    N, H, W, Cn, K, R, S, st, stem = k
    if stem or H == 1:
      continue
    tag = 'C%d-K%d-%dx%d-H%d/%d' % (Cn, K, R, S, H, st)
    if args.only and args.only not in tag:
      continue
    d = ops.make_conv_desc(N, H, W, Cn, K, R, S, st)
    x = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(torch.bfloat16)
    dy = torch.randn((N, d.Ho, d.Wo, K), generator=g, device='cuda').to(torch.bfloat16)
    dw = torch.empty((K, R, S, Cn), dtype=torch.float32, device='cuda')
    fn = lambda: ops.conv_wgrad(d, x, dy, dw)
    set_field('wgrad_splits', 0)
    pl = plan_of(d)
    if pl[1] < 0:        # the halo form has no split knob
      continue
    res = {0: time_fn(fn, args.iters)}
    for sp in splits:
      set_field('wgrad_splits', sp)
      if plan_of(d)[4] != sp and sp != 1:
        continue          # clamped (fewer steps than splits)
      res[sp] = time_fn(fn, args.iters)
    set_field('wgrad_splits', 0)
    res[0] = min(res[0], time_fn(fn, args.iters))     # (the first timing of a shape also pays its workspace and the clock ramp)
    best = min(res, key=res.get)
    rows.append(dict(shape=tag, count=cnt, tile=pl[:2], auto_splits=pl[4], auto_us=round(res[0], 1), best=best,
                     best_us=round(res[best], 1), times={str(s): round(v, 1) for s, v in res.items()}))
    print('%-24s x%d tile %3dx%-3d auto %3d splits %7.1f us | best %3d: %7.1f us | %s' % (
        tag, cnt, pl[0], pl[1], pl[4], res[0], best, res[best],
        ' '.join('%d:%.0f' % (s, v) for s, v in sorted(res.items()) if s)), flush=True)
  a = sum(r['count'] * r['auto_us'] for r in rows)
  b = sum(r['count'] * r['best_us'] for r in rows)
  print('weighted us per step: cost model %.1f, best per shape %.1f' % (a, b))
  if args.out:
    json.dump(rows, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
  main()
