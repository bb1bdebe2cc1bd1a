#!/usr/bin/env python
"""Throughput of the rows either side of the training step: the GPU input-pipeline tail (resize/crop/flip/mean-sub
kernel) and evaluation-mode inference with the on-device metrics (BASELINE configs 1 and 3-eval)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from assembled_cnn_amd import input_pipeline as P, ops
from assembled_cnn_amd.train import HParams, Trainer


def ev(fn, iters=10):
  fn(); torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / iters


rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(int(rng.integers(300, 600)), int(rng.integers(300, 600)), 3), dtype=np.uint8) for _ in range(256)]
for training, side, ptype in ((True, 224, 'imagenet'), (False, 256, 'imagenet_224_256')):
  wins = [P.train_window(im.shape[0], im.shape[1], side, side, rng) if training else P.eval_window(im.shape[0], im.shape[1], side, side)
          for im in imgs]
  buf, table = P.pack_batch(imgs, wins, side, side)
  bd, td = buf.cuda(), table.cuda()
  ms = ev(lambda: ops.resize_crop_flip(bd, td, len(imgs), side, side, True))
  out_bytes = len(imgs) * side * side * 3 * 4
  t0 = time.time(); P.preprocess_batch(imgs, training, 'cuda', preprocessing_type=ptype, windows=wins); torch.cuda.synchronize()
  print('input tail %-5s %dx%d: kernel %.3f ms for 256 images (%.0f k img/s, %.2f TB/s out+in), host pack+H2D+kernel %.1f ms'
        % ('train' if training else 'eval', side, side, ms, 256 / ms, (out_bytes + buf.numel()) / ms / 1e9, 1e3 * (time.time() - t0)))

for name, kw, side in (('ResNet-50 v1.5 eval 224', dict(resnet_version=1), 224),
                       ('Assemble-ResNet-50 eval 256', dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                                                            anti_alias_filter_size=3), 256)):
  hp = HParams(zero_gamma=True, batch_size=256, **kw)
  tr = Trainer(hp, device='cuda')
  x = torch.randn((256, side, side, 3), device='cuda') * 50
  lab = torch.randint(1, 1001, (256,), dtype=torch.int32, device='cuda')
  tr.model.build((side, side))
  ms = ev(lambda: tr.eval_step(x, lab), iters=5)
  print('%-28s batch 256: %.1f ms -> %.0f img/s (forward with moving statistics + top-1/top-5/ECE accumulation)' % (name, ms, 256e3 / ms))
