#!/usr/bin/env python
"""How long does the host take to ENQUEUE one training step (vs the GPU executing it)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd.train import HParams, Trainer

hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
             zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01, batch_size=256,
             label_smoothing=0.1)
tr = Trainer(hp, device='cuda')
img = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, device='cuda')
lab = torch.randint(1, 1001, (256,), dtype=torch.int32, device='cuda')
for _ in range(3):
  tr.train_step(img, lab)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
  torch.cuda.synchronize()
  a = time.perf_counter()
  tr.train_step(img, lab)
  b = time.perf_counter()
  torch.cuda.synchronize()
  c = time.perf_counter()
  enq.append((b - a, c - a))
print('enqueue ms per step: %.2f   enqueue+drain ms: %.2f' % (1e3 * sum(e[0] for e in enq) / 10, 1e3 * sum(e[1] for e in enq) / 10))
