#!/usr/bin/env python
"""IN-SITU sweep of the 1x1 kernel configurations (csrc/conv_gemm1.hip): one configuration at a time is forced on every 1x1
layer of the training step (asm_tuning.gemm1 = code; layers it does not fit stay on igemm2) and every convolution launch of
a few eager steps -- side streams on, i.e. beside the weight gradients of the other streams, as in production -- is
bracketed by HIP events on its launch stream.  Per (kind, layer) the average in-situ duration under every code -> JSON.
Stand-alone sweeps (tools/gemm1_sweep.py) under-predict what a smaller footprint gains beside other streams' kernels by
2.5 x (profiles/round5_gemm1_sweep.md); this measures where the kernels actually run.
usage: insitu_sweep.py [--codes 0,1,5,...] [--steps 5] [--out gpurun_out/insitu.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from assembled_cnn_amd import lib as _lib, ops  # noqa: E402
from assembled_cnn_amd.train import HParams, Trainer  # noqa: E402


def set_field(field, value):
  t = _lib.Tuning()
  ops.L().asm_get_tuning(C.byref(t))
  setattr(t, field, value)
  assert ops.L().asm_set_tuning(C.byref(t)) == 0, ops.L().asm_last_error()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--codes', default='-1,0,1,5,8,10,11,12,13,14,15,16')
  ap.add_argument('--steps', type=int, default=5)
  ap.add_argument('--batch', type=int, default=256)
  ap.add_argument('--workload', default='assemble-r50')
  ap.add_argument('--out', default='gpurun_out/insitu.json')
  ap.add_argument('--settings', default='',
                  help='instead of --codes: "name:field=value,field=value;name2:..." -- asm_tuning fields forced per run; every '
                       'convolution launch is recorded (3x3 and weight gradients too)')
  args = ap.parse_args()
  B = args.batch
  hp = HParams(**dict(dict(resnet_size=50, zero_gamma=True, weight_decay=1e-4, momentum=0.9, base_learning_rate=0.1,
                           learning_rate_decay_type='fixed', batch_size=B, dtype='bf16'), **bench.WORKLOADS[args.workload]['hp']))
  tr = Trainer(hp, seed=0, device='cuda', recorded=False)
  g = torch.Generator(device='cuda').manual_seed(1)
  images = torch.randint(0, 256, (B, 224, 224, 3), generator=g, device='cuda', dtype=torch.uint8)
  labels = torch.randint(1, 1001, (B,), generator=g, device='cuda', dtype=torch.int32)
  for _ in range(3):
    tr.train_step(images, labels)
  res = {}
  if args.settings:
    runs = []
    for spec in args.settings.split(';'):
      nm, _, fv = spec.partition(':')
      runs.append((nm, [(f.split('=')[0], int(f.split('=')[1])) for f in fv.split(',') if f]))
  else:
    runs = [(str(int(c)), [('gemm1', int(c))]) for c in args.codes.split(',')]
  dflt = _lib.Tuning()
  ops.L().asm_tuning_defaults(C.byref(dflt))
  for code, fields in runs:
    assert ops.L().asm_set_tuning(C.byref(dflt)) == 0
    for f, v in fields:
      set_field(f, v)
    for _ in range(2):
      tr.train_step(images, labels)
    torch.cuda.synchronize()
    t = ops.ConvTimer()
    t.split_addend = True
    ops.set_conv_timer(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
      tr.train_step(images, labels)
    e1.record()
    torch.cuda.synchronize()
    ops.set_conv_timer(None)
    summ = t.summary()
    res[str(code)] = {'step_ms': e0.elapsed_time(e1) / args.steps,
                      'convs': {'|'.join(map(str, k)): [n / args.steps, ms / args.steps] for k, (n, ms) in summ.items()
                                if args.settings or (k[6] == 1 and k[7] == 1 and k[2] > 1 and k[0] != 'wgrad')}}
    tot = sum(v[1] for v in res[str(code)]['convs'].values())
    print('%-8s: instrumented eager step %.2f ms, recorded convolutions in situ %.3f ms' % (code, res[str(code)]['step_ms'], tot), flush=True)
  assert ops.L().asm_set_tuning(C.byref(dflt)) == 0
  os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
  json.dump(res, open(args.out, 'w'))
  if args.settings:
    return
  # per-layer best
  keys = sorted(res['0']['convs'])
  t0 = tb = 0.0
  for k in keys:
    base = res['0']['convs'][k][1]
    best = min((c for c in res if c not in ('-1',) and k in res[c]['convs']), key=lambda c: res[c]['convs'][k][1])
    t0 += base
    tb += res[best]['convs'][k][1]
  print('1x1 fprop + dgrad in situ: all igemm2 %.3f ms, table of this build %.3f ms, best code per layer %.3f ms' % (
      t0, sum(v[1] for v in res['-1']['convs'].values()) if '-1' in res else float('nan'), tb))


if __name__ == '__main__':
  main()
