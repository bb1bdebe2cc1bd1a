#!/usr/bin/env python
"""Race hunt for the ring / stagger weight-gradient kernels: every launch of a layer must reproduce the first one BIT FOR BIT
(fixed-order slab reduce), also while another stream keeps the chip busy with bandwidth-bound kernels and the inputs are
re-randomised between layers.  usage (GPU box): python tools/debug/wgrad_stress.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd import ops  # noqa: E402

SHAPES = [  # N, H, W, C, K, k, stride
    (256, 14, 14, 512, 1024, 3, 1), (256, 7, 7, 512, 1024, 3, 1), (256, 28, 28, 128, 256, 3, 1), (256, 14, 14, 1024, 1024, 1, 1),
    (256, 28, 28, 256, 512, 1, 1), (256, 56, 56, 64, 128, 3, 1), (256, 28, 28, 64, 128, 3, 1), (256, 56, 56, 32, 64, 3, 1),
    (256, 112, 112, 64, 32, 3, 1), (64, 14, 14, 256, 256, 3, 2), (256, 7, 7, 256, 512, 3, 1)]


def main():
  iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
  g = torch.Generator(device='cuda').manual_seed(7)
  side = torch.cuda.Stream()
  noise = torch.randn((64 << 20,), device='cuda', generator=g)
  bad = 0
  for (N, H, W, Cn, K, k, st) in SHAPES:
    d = ops.make_conv_desc(N, H, W, Cn, K, k, k, st)
    x = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(torch.bfloat16)
    dy = torch.randn((N, d.Ho, d.Wo, K), generator=g, device='cuda').to(torch.bfloat16)
    ref = torch.empty((K, k, k, Cn), dtype=torch.float32, device='cuda')
    ops.conv_wgrad(d, x, dy, ref)
    want = None
    if N * H * W <= 60000:     # an fp64 product for the small ones
      pass
    torch.cuda.synchronize()
    diff = 0
    for it in range(iters):
      if it % 2:
        with torch.cuda.stream(side):      # a bandwidth-bound neighbour on another stream
          noise.mul_(1.0001)
      dw = torch.full_like(ref, float('nan'))
      ops.conv_wgrad(d, x, dy, dw)
      if not torch.equal(dw, ref):
        diff += 1
    torch.cuda.synchronize()
    print('%-28s %d launches, %d differ from the first' % ('x'.join(map(str, (N, H, W, Cn, K, k, st))), iters, diff), flush=True)
    bad += diff
  print('TOTAL differing launches:', bad)
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
