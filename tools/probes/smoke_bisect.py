"""Which step of __graft_entry__.smoke() makes the next launch of this library fail?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g


def probe(tag):
  from assembled_cnn_amd import ops
  try:
    a = torch.ones(16, device='cuda'); b = torch.empty(16, dtype=torch.bfloat16, device='cuda')
    ops.cast_f32_to_bf16(a, b); torch.cuda.synchronize()
    print(tag, 'ok', flush=True)
  except Exception as e:
    print(tag, 'FAIL', e, flush=True)


g.build(); probe('after build')
assert torch.cuda.is_available(); probe('after is_available')
from tests import model_parity as mp; probe('after import model_parity')
from oracle import assembled_oracle as O
kw = mp.CONFIGS['a-r50-d']
om = O.Model(num_classes=1001, emulate_bf16=True, zero_gamma=True, seed=0, **kw); probe('after oracle ctor')
om(torch.zeros(2, 64, 64, 3), True, use_resnet_d=True); probe('after oracle forward')
from assembled_cnn_amd.model import Model
pm = Model(num_classes=1001, device='cuda', zero_gamma=True, seed=0, **kw); probe('after product ctor')
pm.build((64, 64), use_resnet_d=True); probe('after product build')
