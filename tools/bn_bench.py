#!/usr/bin/env python
"""Micro-benchmark of the batch-norm kernels on the workload's tensor shapes (GPU only): effective TB/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd import ops
from assembled_cnn_amd.ops import L, _ptr, _stream, check

SHAPES = [(256 * 112 * 112, 64), (256 * 112 * 112, 32), (256 * 56 * 56, 128), (256 * 56 * 56, 256), (256 * 56 * 56, 64),
          (256 * 28 * 28, 512), (256 * 28 * 28, 256), (256 * 14 * 14, 1024), (256 * 14 * 14, 512), (256 * 7 * 7, 2048),
          (256 * 7 * 7, 512)]


def timeit(fn, iters=10):
  fn(); fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e3


def main():
  g = torch.Generator(device='cuda').manual_seed(0)
  print('%-18s | %-22s | %-22s | %-22s | %-22s' % ('M x C', 'bwd_reduce (4B/el)', 'bwd_apply (6B/el)', 'apply+relu (4B/el)', 'apply+res+relu (6B)'))
  tot = [0.0] * 4
  for M, Cn in SHAPES:
    x = torch.randn((M, Cn), generator=g, device='cuda').to(torch.bfloat16)
    dy = torch.randn((M, Cn), generator=g, device='cuda').to(torch.bfloat16)
    res = torch.randn((M, Cn), generator=g, device='cuda').to(torch.bfloat16)
    gamma = torch.ones(Cn, device='cuda'); beta = torch.zeros(Cn, device='cuda')
    part = ops.bn_stats(x, M, Cn)
    mean, invstd, scale, shift = ops.bn_finalize(part, M, Cn, gamma, beta, 1e-5, 0.997, None, None)
    y, mask = ops.bn_apply(x, M, Cn, scale, shift, relu=True, want_mask=True)
    blocks = L().asm_bn_stats_blocks(M, Cn)
    p2 = torch.empty((blocks, 2, Cn), device='cuda')
    co = torch.randn((3, Cn), device='cuda')
    dx = torch.empty_like(x)
    t_red = timeit(lambda: check(L().asm_bn_bwd_reduce(_ptr(dy), _ptr(x), _ptr(mask), 2, M, Cn, _ptr(mean), _ptr(invstd), _ptr(p2), _stream()), 'r'))
    t_app = timeit(lambda: check(L().asm_bn_bwd_apply(_ptr(dy), _ptr(x), _ptr(mask), 2, M, Cn, _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _ptr(dx), None, _stream()), 'a'))
    t_fwd = timeit(lambda: ops.bn_apply(x, M, Cn, scale, shift, relu=True, want_mask=True))
    t_res = timeit(lambda: ops.bn_apply(x, M, Cn, scale, shift, residual=res, res_mode=1, relu=True, want_mask=True))
    n = M * Cn
    tb = lambda bytes_, us: bytes_ / us / 1e6
    print('%9d x %-6d | %7.1f us %5.2f TB/s | %7.1f us %5.2f TB/s | %7.1f us %5.2f TB/s | %7.1f us %5.2f TB/s' % (
        M, Cn, t_red, tb(4.125 * n, t_red), t_app, tb(6.125 * n, t_app), t_fwd, tb(4.125 * n, t_fwd), t_res, tb(6.125 * n, t_res)))
    for i, t in enumerate((t_red, t_app, t_fwd, t_res)):
      tot[i] += t
  print('sum us:', ['%.0f' % t for t in tot])


if __name__ == '__main__':
  main()
