"""Topology self-consistency pins of the oracle: parameter / trainable-tensor counts the
reference's graph must have, output shapes, loss at init."""
import math

import pytest
import torch

from oracle import assembled_oracle as O

AR50 = dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3)
CASES = [
    ('r50v1', dict(resnet_size=50), False, 25559081, 161),
    ('a-r50', AR50, False, 41848489, 306),
    ('a-r50+d', AR50, True, 41867721, 312),
    ('a-r152', dict(AR50, resnet_size=152, bl_alpha=1, bl_beta=2), False, 117006249, 969),
]


@pytest.mark.parametrize('name,kw,d,params,tensors', CASES, ids=[c[0] for c in CASES])
def test_param_and_tensor_counts(name, kw, d, params, tensors):
  m = O.Model(num_classes=1001, **kw)
  y = m(torch.zeros(1, 64, 64, 3), False, use_resnet_d=d)
  assert y.shape == (1, 1001)
  assert m.vars.num_params() == params
  assert len(m.vars.trainable) == tensors
  decayed = [n for n in m.vars.trainable if 'batch_normalization' not in n]
  assert any(n.endswith('dense/bias') for n in decayed), 'dense bias is weight-decayed (Appendix A.10)'


def test_r50_canonical_param_count_without_background_class():
  m = O.Model(50, num_classes=1000)
  m(torch.zeros(1, 64, 64, 3), False)
  assert m.vars.num_params() == 25557032  # the canonical ResNet-50 v1.5 figure


def test_stage_shapes_and_taps():
  m = O.Model(50, num_classes=1001)
  m(torch.zeros(2, 224, 224, 3), False)
  t = m.taps_nhwc()
  assert t['initial_conv'].shape == (2, 112, 112, 64)
  assert t['initial_max_pool'].shape == (2, 56, 56, 64)
  assert [tuple(t['block_layer%d' % i].shape[1:]) for i in (1, 2, 3, 4)] == [
      (56, 56, 256), (28, 28, 512), (14, 14, 1024), (7, 7, 2048)]
  a = O.Model(num_classes=1001, **AR50)
  a(torch.zeros(1, 224, 224, 3), False)
  t = a.taps_nhwc()
  assert [tuple(t[k].shape[1:]) for k in ('merge1', 'merge2', 'merge3', 'block_layer4')] == [
      (28, 28, 256), (14, 14, 512), (14, 14, 1024), (7, 7, 2048)]
  assert tuple(t['big1'].shape[1:]) == (28, 28, 256) and tuple(t['little1'].shape[1:]) == (56, 56, 128)


def test_loss_at_init_is_log_num_classes_with_zero_gamma():
  torch.manual_seed(0)
  m = O.Model(50, num_classes=1001, zero_gamma=True)
  x = torch.randn(4, 64, 64, 3) * 50
  logits = m(x, True)
  onehot = torch.nn.functional.one_hot(torch.tensor([1, 2, 3, 4]), 1001).float()
  ce = float(O.softmax_cross_entropy(logits, onehot))
  assert abs(ce - math.log(1001)) < 0.5


def test_errors_mirror_reference():
  with pytest.raises(ValueError):
    O.Model(50, num_classes=10, resnet_version=3)
  with pytest.raises(ValueError):
    O.Model(77, num_classes=10)
  with pytest.raises(NotImplementedError):
    O.Model(18, num_classes=10)
  with pytest.raises(NotImplementedError):
    O.Model(50, num_classes=10, pool_type='foo')
