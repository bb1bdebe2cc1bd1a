#!/usr/bin/env python
"""cProfile of the host side of the training step (where do the ~18 ms of enqueue time per step go?)"""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd.train import HParams, Trainer

hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
             zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01, batch_size=256)
tr = Trainer(hp, device='cuda')
img = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, device='cuda')
lab = torch.randint(1, 1001, (256,), dtype=torch.int32, device='cuda')
for _ in range(3):
  tr.train_step(img, lab)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
  tr.train_step(img, lab)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
