#!/usr/bin/env python
"""List every conv launch (shape, count, GFLOP) of a workload at a given batch: walks the model in its
shape-only build mode with ops.make_conv_desc instrumented.  usage: list_convs.py [workload] [batch]"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from assembled_cnn_amd import nn, ops  # noqa: E402
from assembled_cnn_amd.train import HParams  # noqa: E402


def conv_shapes(workload='assemble-r50', batch=256):
  hp = HParams(**dict(dict(resnet_size=50, zero_gamma=True), **bench.WORKLOADS[workload]['hp']))
  m = hp.make_model(device='cpu')
  seen = OrderedDict()
  orig = nn.ConvKernel.desc

  def desc(self, N, H, W, stride, out_f32=False, ldy=0):
    d = orig(self, N, H, W, stride, out_f32, ldy)
    key = (d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride, bool(self.stem))
    seen[key] = seen.get(key, 0) + 1
    return d
  nn.ConvKernel.desc = desc
  ctx = nn.Ctx(m.arena, True, True, 0.997, 'cpu', False, m._layers)
  m._walk(ctx, nn.Var(None, (batch, 230, 230, 4), needs_grad=False), hp.use_resnet_d, False)
  nn.ConvKernel.desc = orig
  return seen


if __name__ == '__main__':
  wl = sys.argv[1] if len(sys.argv) > 1 else 'assemble-r50'
  b = int(sys.argv[2]) if len(sys.argv) > 2 else 256
  tot = 0.0
  rows = []
  for k, cnt in conv_shapes(wl, b).items():
    N, H, W, Cn, K, R, S, st, stem = k
    Ho = H if st == 1 else (H - 1) // st + 1
    Wo = W if st == 1 else (W - 1) // st + 1
    if stem:
      Ho, Wo = (H - 6 - 1) // 2 + 1, (W - 6 - 1) // 2 + 1
    gf = 2.0 * N * Ho * Wo * K * Cn * R * S / 1e9
    rows.append((gf * cnt, cnt, k, Ho, gf))
    tot += gf * cnt
  rows.sort(reverse=True)
  print('%d shapes, fwd GFLOP %.1f (x3 for train)' % (len(rows), tot))
  for g, cnt, k, Ho, gf in rows:
    print('%8.1f GF x%d  N%d %dx%d C%d -> K%d %dx%d/%d%s  (M=%d)' % (gf, cnt, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7],
                                                                   ' stem' if k[8] else '', k[0] * Ho * Ho))
