#!/usr/bin/env python
"""Which variables does an exchange at world size 1 change?  MODE=bf16: gradients vs bf16(plain gradients);
MODE=fp32: weights after whole steps vs the plain trainer's (see tests/test_gpu_dp_rccl.py)"""
import os, socket, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd import dp
from assembled_cnn_amd.train import HParams, Trainer
from tests import model_parity as mpar

MODE = os.environ.get('MODE', 'bf16')
B, S = int(os.environ.get('BATCH', '32')), int(os.environ.get('SIZE', '128'))
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
             zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01, weight_decay=1e-4, batch_size=B)
img, _, labels = mpar.inputs(B, S)
img, labels = img.cuda(), labels.cuda()
plain = Trainer(hp, seed=0, device='cuda'); plain.model.build((S, S), use_resnet_d=True)
tr = Trainer(hp, seed=0, device='cuda', world_size=1); tr.model.build((S, S), use_resnet_d=True)
sync = dp.GradSync(tr.model.arena, bucket_bytes=int(os.environ.get('BUCKET_MB', '4')) << 20, comm_dtype=MODE)


def report(want, got, what):
  a = plain.model.arena
  print('   %s: mismatching elements %d' % (what, int((want != got).sum())))
  n = 0
  for name, sp in a.specs.items():
    w, g = want[sp.offset:sp.offset + sp.numel], got[sp.offset:sp.offset + sp.numel]
    bad = int((w != g).sum())
    if bad:
      n += 1
      if n <= 25:
        print('   %-70s off %9d  bad %8d / %8d  max|want| %.3e max|got| %.3e' % (name, sp.offset, bad, sp.numel, float(w.abs().max()), float(g.abs().max())))
  print('   variables with mismatches:', n, 'of', len(a.specs))


for step in range(int(os.environ.get('STEPS', '4'))):
  print('step', step)
  if MODE == 'bf16':
    plain._forward_backward(img, labels, None, None)
    tr._forward_backward(img, labels, None, None)
    sync(tr.model.arena.g32)
    torch.cuda.synchronize()
    report(plain.model.arena.g32.to(torch.bfloat16).float(), tr.model.arena.g32, 'gradients')
    tr.model.arena.g32.copy_(plain.model.arena.g32)
    plain._apply(None, 1.0, 1.0); tr.grad_sync = None; tr._apply(None, 1.0, 1.0)
  else:
    tr.grad_sync = sync
    plain.train_step(img, labels)
    tr.train_step(img, labels)
    torch.cuda.synchronize()
    report(plain.model.arena.w32, tr.model.arena.w32, 'weights')
dist.destroy_process_group()
