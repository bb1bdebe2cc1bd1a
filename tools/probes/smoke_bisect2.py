"""first launch of the library only AFTER the oracle's CPU forward (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'

def probe(tag):
  from assembled_cnn_amd import ops
  try:
    a = torch.ones(16, device='cuda'); b = torch.empty(16, dtype=torch.bfloat16, device='cuda')
    ops.cast_f32_to_bf16(a, b); torch.cuda.synchronize()
    print(tag, 'ok', flush=True)
  except Exception as e:
    print(tag, 'FAIL', e, flush=True)

g.build()
assert torch.cuda.is_available()
if mode == 'touch':
  torch.ones(4, device='cuda').sum().item()       # torch initialises its context first, no launch of ours
from tests import model_parity as mp
from oracle import assembled_oracle as O
kw = mp.CONFIGS['a-r50-d']
om = O.Model(num_classes=1001, emulate_bf16=True, zero_gamma=True, seed=0, **kw)
if mode != 'noforward':
  om(torch.zeros(2, 64, 64, 3), True, use_resnet_d=True)
print('threads', torch.get_num_threads(), flush=True)
probe('first launch, mode ' + mode)
probe('second launch')
