"""smoke()'s exact flow with an optional action right before the product build (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g
mode = sys.argv[1]
g.build()
assert torch.cuda.is_available()
from tests import model_parity as mp
from assembled_cnn_amd import lib, ops
ops.set_library(None, is_double=False)
lib.load()
if mode == 'warm':
  a = torch.ones(16, device='cuda'); b = torch.empty(16, dtype=torch.bfloat16, device='cuda')
  ops.cast_f32_to_bf16(a, b)
elif mode == 'sync':
  torch.cuda.synchronize()
elif mode == 'bigcast':
  a = torch.ones(41900000, device='cuda'); b = torch.empty(41900000, dtype=torch.bfloat16, device='cuda')
  try:
    ops.cast_f32_to_bf16(a, b); torch.cuda.synchronize(); print('bigcast ok')
  except Exception as e:
    print('bigcast FAIL', e)
try:
  e = mp.check_forward('a-r50-d', 'cuda', 8, 64, True, 6e-2)
  print(mode, 'ok', e, flush=True)
except Exception as ex:
  print(mode, 'FAIL', repr(ex)[:200], flush=True)
