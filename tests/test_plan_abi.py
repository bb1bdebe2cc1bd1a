"""asm_model_cfg / asm_model_plan (include/asm_hip.h): the topology through the C ABI, without a GPU.

The planner's entries are expanded to TensorFlow variable names and shapes and must equal, name for name and in
creation order, the variables the REFERENCE'S OWN CODE created under the tf shim (tests/golden/reference_taps.json),
for all 8 fixture configurations; parameter totals equal the published / SURVEY pins; errors follow the reference's
ValueError / NotImplementedError split."""
import ctypes as C
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, 'golden', 'reference_taps.json')))

HP = {
    'r50v1': dict(),
    'r50v1-d': dict(use_resnet_d=True),
    'a-r50': dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3),
    'a-r50-d': dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True),
    'a-r152': dict(resnet_size=152, resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                   bl_alpha=1, bl_beta=2),
    'se-proj': dict(use_se_block=True, anti_alias_type='proj', anti_alias_filter_size=3),
    'r101v1-gem-emb': dict(resnet_size=101, pool_type='gem', embedding_size=128, zero_gamma=True),
    'r50v1-nodown-flatten-sigmoid': dict(no_downsample=True, pool_type='flatten', cls_loss_type='sigmoid'),
}


def _plan(hp, N, H, W):
  from assembled_cnn_amd import lib
  L = lib.load()
  cfg = hp.to_cfg()
  summ = lib.PlanSummary()
  rc = L.asm_model_plan(C.byref(cfg), N, H, W, None, 0, C.byref(summ))
  lib.check(rc, 'model_plan')
  ent = (lib.PlanEntry * summ.n_entries)()
  lib.check(L.asm_model_plan(C.byref(cfg), N, H, W, ent, summ.n_entries, C.byref(summ)), 'model_plan')
  return list(ent), summ


def _expand(entries):
  from assembled_cnn_amd import lib
  out = []
  for e in entries:
    name = e.name.decode()
    if e.kind == lib.PLAN_CONV:
      out.append([name + '/kernel', [e.R, e.S, e.C, e.K], True])
    elif e.kind == lib.PLAN_BN:
      out += [[name + '/gamma', [e.C], True], [name + '/beta', [e.C], True],
              [name + '/moving_mean', [e.C], False], [name + '/moving_variance', [e.C], False]]
    elif e.kind == lib.PLAN_DENSE:
      out += [[name + '/kernel', [e.C, e.K], True], [name + '/bias', [e.K], True]]
  return out


@pytest.mark.parametrize('name', sorted(HP))
def test_plan_variable_table_equals_reference_graph(name):
  from assembled_cnn_amd import lib
  from assembled_cnn_amd.train import HParams
  fx = FIX['models'][name]
  hp = HParams(**dict(dict(resnet_size=50), **HP[name]))
  size = fx['input_size']
  entries, summ = _plan(hp, 2, size, size)
  assert _expand(entries) == fx['variables'], 'planner variables differ from the reference graph'
  n_train = sum(1 for _, _, tr in fx['variables'] if tr)
  assert summ.trainable_tensors == n_train
  elems = 0
  for n, shp, tr in fx['variables']:
    if tr:
      k = 1
      for d in shp:
        k *= d
      elems += k
  assert summ.trainable_elems == elems
  # offsets are the running sum in creation order
  run = 0
  for e in entries:
    if e.trainable:
      assert e.param_offset == run
      run += e.param_elems
  zg = [e.name.decode() + '/gamma' for e in entries if e.kind == lib.PLAN_BN and e.flags & 8]
  assert zg == fx['zero_gammas']


def test_plan_totals_and_shapes_at_224():
  """SURVEY 8c pins: parameters 25 559 081 / 41 848 489 / 41 867 721 / 117 006 249, tensors 161 / 306 / 312 / 969,
  forward 4.089 GMAC for ResNet-50 v1.5; and the product walker's conv descriptors at batch 256."""
  import sys
  from assembled_cnn_amd import lib
  from assembled_cnn_amd.train import HParams
  want = {'r50v1': (25559081, 161), 'a-r50': (41848489, 306), 'a-r50-d': (41867721, 312), 'a-r152': (117006249, 969)}
  for name, (params, tensors) in want.items():
    entries, summ = _plan(HParams(**dict(dict(resnet_size=50), **HP[name])), 256, 224, 224)
    assert (summ.trainable_elems, summ.trainable_tensors) == (params, tensors), name
    if name == 'r50v1':
      assert abs(summ.forward_macs_per_image / 1e9 - 4.089) < 0.01
    assert summ.wgrad_workspace_bytes > 0
  sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tools'))
  import list_convs
  entries, _ = _plan(HParams(**dict(dict(resnet_size=50), **HP['a-r50-d'])), 256, 224, 224)
  # (the walker's shape-only pass does not descend into the [N,1,1,d] squeeze convs of the SK units)
  mine = [(e.H, e.W, e.C, e.K, e.R, e.S, e.stride) for e in entries if e.kind == lib.PLAN_CONV and e.C != 3 and e.H * e.W > 1]
  walker = set((k[1], k[2], k[3], k[4], k[5], k[6], k[7]) for k in list_convs.conv_shapes('assemble-r50', 256) if not k[8])
  assert set(mine) == walker


def test_plan_errors_follow_the_reference():
  from assembled_cnn_amd import lib
  from assembled_cnn_amd.train import HParams
  with pytest.raises(ValueError):
    _plan(HParams(resnet_size=50, resnet_version=3), 2, 64, 64)            # nets/resnet_model.py:200-203
  with pytest.raises(ValueError):
    _plan(HParams(resnet_size=77), 2, 64, 64)                             # functions/model_fns.py:131-135
  with pytest.raises(NotImplementedError):
    _plan(HParams(resnet_size=34), 2, 64, 64)                             # non-bottleneck
  with pytest.raises(NotImplementedError):
    _plan(HParams(resnet_size=50, dtype='fp16'), 2, 64, 64)               # reference dtype not computed here
  with pytest.raises(ValueError):
    _plan(HParams(resnet_size=200, resnet_version=2), 2, 64, 64)          # no BigLittle ResNet-200
