#!/usr/bin/env python
"""The step as one HIP graph: does capture work, is it bit-identical to the eager step, what does it cost per step?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd.train import HParams, Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hp = dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
          zero_gamma=True, learning_rate_decay_type='cosine', base_learning_rate=0.01, batch_size=B, label_smoothing=0.1)
g = torch.Generator(device='cuda').manual_seed(3)
imgs = [torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device='cuda', generator=g) for _ in range(2)]
labs = [torch.randint(1, 1001, (B,), dtype=torch.int32, device='cuda', generator=g) for _ in range(2)]


def run(graphed, steps=6):
  tr = Trainer(HParams(**hp), seed=0, device='cuda')
  if graphed:
    tr.capture(imgs[0], labs[0], warmup=0) if False else None
  losses = []
  for s in range(steps):
    if graphed and s == 2:
      tr.capture(imgs[0], labs[0], warmup=0)
    tr.train_step(imgs[s % 2], labs[s % 2])
    losses.append(float(tr.cross_entropy()))
  torch.cuda.synchronize()
  return tr, losses


te, le = run(False)
tg, lg = run(True)
print('eager  ', le)
print('graphed', lg)
same = torch.equal(te.model.arena.w32, tg.model.arena.w32) and torch.equal(te.model.arena.m32, tg.model.arena.m32)
print('weights + momentum bit-identical after 6 steps (2 eager + 4 replayed vs 6 eager):', same)
st = [torch.equal(te.model.arena.st(n), tg.model.arena.st(n)) for n in te.model.arena.state_specs]
print('moving statistics bit-identical:', all(st))
for name, tr in (('eager', te), ('graphed', tg)):
  for _ in range(3):
    tr.train_step(imgs[0], labs[0])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(30):
    tr.train_step(imgs[0], labs[0])
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('%s: %.3f ms per step (host enqueue %.3f ms per step)' % (name, 1e3 * (t2 - t0) / 30, 1e3 * (t1 - t0) / 30))
