#!/usr/bin/env python
"""Per-tile fixed cost of the implicit-GEMM kernels: 1x1 convolutions of a fixed [M x N] output with the reduction length
swept (time = rounds * (steps * a + b)); a = per-step time, b = prologue + epilogue + workgroup turn-around."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd import ops  # noqa: E402


def t_us(fn, iters=50):
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
  g = torch.Generator(device='cuda').manual_seed(0)
  for (N, H, K) in ((256, 14, 1024), (256, 14, 512), (256, 28, 256), (256, 7, 2048)):
    print('== output %d x %d x %d x %d' % (N, H, H, K))
    for Cn in (64, 128, 256, 512, 1024, 2048):
      d = ops.make_conv_desc(N, H, H, Cn, K, 1, 1, 1)
      x = torch.randn((N, H, H, Cn), generator=g, device='cuda').to(torch.bfloat16)
      w = (torch.randn((K, 1, 1, Cn), generator=g, device='cuda') * Cn ** -0.5).to(torch.bfloat16)
      row = []
      for stats in (True, False):
        row.append(t_us(lambda: ops.conv_fprop(d, x, w, stats)))
      print('  C %5d  steps(64) %3d   stats %7.1f us   plain %7.1f us' % (Cn, Cn // 64, row[0], row[1]))


if __name__ == '__main__':
  main()
