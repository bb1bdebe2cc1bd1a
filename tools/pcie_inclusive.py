#!/usr/bin/env python
"""Training throughput when every step's uint8 batch starts in pinned HOST memory (PCIe-inclusive rate):
(a) copy on the compute stream, (b) double-buffered copy on a side stream overlapped with the previous step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd.train import HParams, Trainer

hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
             zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.1, weight_decay=1e-4, batch_size=256)
tr = Trainer(hp, device='cuda')
B = 256
host = [torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
lab = torch.randint(1, 1001, (B,), dtype=torch.int32, device='cuda')
dev = [torch.empty((B, 224, 224, 3), dtype=torch.uint8, device='cuda') for _ in range(2)]
dev[0].copy_(host[0])
for _ in range(5):
  tr.train_step(dev[0], lab)
torch.cuda.synchronize()


def run(mode, steps=20):
  copy_stream = torch.cuda.Stream()
  ready = [torch.cuda.Event(), torch.cuda.Event()]
  torch.cuda.synchronize()
  t0 = time.time()
  if mode == 'overlap':
    with torch.cuda.stream(copy_stream):
      dev[0].copy_(host[0], non_blocking=True); ready[0].record()
  for s in range(steps):
    cur = s & 1
    if mode == 'resident':
      tr.train_step(dev[0], lab)
    elif mode == 'inline':
      dev[0].copy_(host[cur], non_blocking=True)
      tr.train_step(dev[0], lab)
    else:
      torch.cuda.current_stream().wait_event(ready[cur])
      with torch.cuda.stream(copy_stream):            # next batch while this step computes
        copy_stream.wait_stream(torch.cuda.current_stream()) if s else None
        dev[cur ^ 1].copy_(host[cur ^ 1], non_blocking=True); ready[cur ^ 1].record()
      tr.train_step(dev[cur], lab)
  torch.cuda.synchronize()
  el = time.time() - t0
  print('%-9s %8.1f img/s  %6.2f ms/step' % (mode, B * steps / el, 1e3 * el / steps))


for m in ('resident', 'inline', 'overlap', 'resident'):
  run(m)
