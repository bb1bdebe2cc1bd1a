"""Known-answer tests pinning every TensorFlow-semantics rule the oracle restates.
TF cannot arbitrate here (not installable), so each rule is checked on a tiny hand-computed example."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import assembled_oracle as O


def _t(a):
  return torch.tensor(a, dtype=torch.float32)


def test_fixed_padding_split():
  x = torch.ones(1, 1, 2, 2)
  assert O.fixed_padding(x, 3).shape == (1, 1, 4, 4)          # 1 before, 1 after
  p = O.fixed_padding(x, 2)                                    # 0 before, 1 after
  assert p.shape == (1, 1, 3, 3) and float(p[0, 0, 0, 0]) == 1 and float(p[0, 0, 2, 2]) == 0
  p7 = O.fixed_padding(x, 7)
  assert p7.shape == (1, 1, 8, 8) and float(p7[0, 0, 3, 3]) == 1 and float(p7[0, 0, 2, 2]) == 0


def test_conv_stride2_is_pad_before_then_valid():
  # 1-D intuition on a 4x4 ramp, 3x3 all-ones filter, stride 2: windows start at -1 and 1
  x = torch.arange(16.).view(1, 1, 4, 4)
  w = torch.ones(3, 3, 1, 1)
  y = O._conv_raw(x, w, 3, 2)
  assert y.shape == (1, 1, 2, 2)
  assert float(y[0, 0, 0, 0]) == 0 + 1 + 4 + 5                # rows -1..1, cols -1..1
  assert float(y[0, 0, 1, 1]) == sum([5, 6, 7, 9, 10, 11, 13, 14, 15])
  # stride 1 keeps the size (SAME)
  assert O._conv_raw(x, w, 3, 1).shape == (1, 1, 4, 4)
  # output size formula for odd inputs
  assert O._conv_raw(torch.zeros(1, 1, 7, 7), w, 3, 2).shape == (1, 1, 4, 4)


def test_maxpool_same_pads_after_only():
  # 4 wide, k3 s2 SAME: out=2, pad_total=1 -> 0 before / 1 after: windows [0,1,2] and [2,3,pad]
  x = _t([[1, 9, 2, 3], [0, 0, 0, 0], [0, 0, 0, 0], [5, 0, 0, 7]]).view(1, 1, 4, 4)
  y = O.max_pool_same(x, 3, 2)
  assert y.shape == (1, 1, 2, 2)
  assert y.view(-1).tolist() == [9, 3, 5, 7]
  # torch's symmetric padding=1 would give a different answer (window [-1,0,1] -> 9, [1,2,3] -> 9)
  assert F.max_pool2d(x, 3, 2, padding=1).view(-1).tolist() != y.view(-1).tolist()
  assert O._same_pad(112, 3, 2) == (56, 0, 1)


def test_avgpool_variants():
  x = torch.arange(16.).view(1, 1, 4, 4)
  # BL shortcut: zero pad 1/1, 3x3/2 VALID, divisor always 9
  bl = O.avg_pool_valid(O.fixed_padding(x, 3), 3, 2)
  assert bl.shape == (1, 1, 2, 2)
  assert abs(float(bl[0, 0, 0, 0]) - (0 + 1 + 4 + 5) / 9) < 1e-6
  # ResNet-D stride 2: pad 0 before / 1 after, 2x2/2 VALID (pad never read for even sizes)
  dd = O.avg_pool_valid(O.fixed_padding(x, 2), 2, 2)
  assert dd.view(-1).tolist() == [2.5, 4.5, 10.5, 12.5]
  # ResNet-D stride 1: 2x2 SAME divides by the number of valid elements
  s1 = O.avg_pool_same(x, 2, 1)
  assert s1.shape == (1, 1, 4, 4)
  assert float(s1[0, 0, 0, 0]) == 2.5                              # (0+1+4+5)/4
  assert float(s1[0, 0, 0, 3]) == (3 + 7) / 2                      # right edge: 2 valid
  assert float(s1[0, 0, 3, 3]) == 15.0                             # corner: 1 valid


def test_upsample_nearest():
  x = _t([[1, 2], [3, 4]]).view(1, 1, 2, 2)
  assert O.upsample2x_nearest(x)[0, 0].tolist() == [[1, 1, 2, 2], [1, 1, 2, 2], [3, 3, 4, 4], [3, 3, 4, 4]]


def test_blur_filter_and_reflect_pad():
  f = O.blur_filter(3)
  assert abs(float(f.sum()) - 1) < 1e-7 and float(f[1, 1]) == 0.25 and float(f[0, 0]) == 1 / 16
  for k in range(1, 8):
    assert abs(float(O.blur_filter(k).sum()) - 1) < 1e-6
    assert torch.equal(O.blur_filter(k, torch.bfloat16).float(), O.blur_filter(k))  # dyadic: exact in bf16
  x = torch.arange(16.).view(1, 1, 4, 4)
  ctx = O.Ctx(O.VarStore(0), False)
  y = O.anti_aliased_downsample(ctx, x, 3, 2)
  assert y.shape == (1, 1, 2, 2)
  # REFLECT (no edge repeat): index -1 -> 1.  top-left window rows/cols (1,0,1)
  rows = [1, 0, 1]
  exp = sum(f[i, j] * x[0, 0, rows[i], rows[j]] for i in range(3) for j in range(3))
  assert abs(float(y[0, 0, 0, 0]) - float(exp)) < 1e-6
  # a constant image stays constant
  assert torch.allclose(O.anti_aliased_downsample(ctx, torch.full((1, 2, 6, 6), 3.0), 3, 2), torch.full((1, 2, 3, 3), 3.0))


def test_batch_norm_train_and_moving_stats():
  vs = O.VarStore(0)
  ctx = O.Ctx(vs, False)
  vs.begin_call()
  x = _t([1., 2., 3., 6.]).view(4, 1, 1, 1)
  y = O.batch_norm(ctx, x, True, momentum=0.9)
  mean, var = 3.0, (4 + 1 + 0 + 9) / 4.0                       # biased variance 3.5
  assert torch.allclose(y.view(-1), (x.view(-1) - mean) / math.sqrt(var + 1e-5), atol=1e-6)
  mm = vs.pending_updates['resnet_model/batch_normalization/moving_mean']
  mv = vs.pending_updates['resnet_model/batch_normalization/moving_variance']
  assert abs(float(mm) - (0 * 0.9 + 3.0 * 0.1)) < 1e-6             # momentum weights the OLD value
  assert abs(float(mv) - (1 * 0.9 + (14 / 3.0) * 0.1)) < 1e-6      # Bessel-corrected (n-1) for the moving var
  vs.apply_updates()
  vs.begin_call()
  y2 = O.batch_norm(ctx, x, False, momentum=0.9)                   # eval: uses the moving stats
  assert torch.allclose(y2.view(-1), (x.view(-1) - 0.3) / math.sqrt(float(mv) + 1e-5), atol=1e-6)


def test_zero_gamma_only_on_block_final_bn():
  m = O.Model(50, num_classes=1001, zero_gamma=True)
  m(torch.zeros(1, 64, 64, 3), False)
  zeros = [n for n, t in m.vars.trainable.items() if n.endswith('gamma') and float(t.abs().sum()) == 0]
  assert len(zeros) == 16                                           # one per bottleneck block, none on shortcuts


def test_sk_gates_and_shapes():
  vs = O.VarStore(0)
  ctx = O.Ctx(vs, False)
  vs.begin_call()
  x = torch.randn(3, 16, 8, 8)
  v = O.sk_conv2d(ctx, x, 16, 1, True)
  assert v.shape == (3, 16, 8, 8)
  names = list(vs.trainable.keys())
  assert names[0].endswith('conv2d/kernel') and vs.trainable[names[0]].shape == (3, 3, 16, 32)   # ONE conv to 2F
  assert vs.trainable['resnet_model/sk_block/sk_fc_1/kernel'].shape == (1, 1, 16, 32)             # d = max(F/2, 32)
  assert vs.trainable['resnet_model/sk_block/sk_fc_2/kernel'].shape == (1, 1, 32, 32)
  # with fc2 == 0 both gates are 1/2 -> V = (f0 + f1) / 2
  with torch.no_grad():
    vs.trainable['resnet_model/sk_block/sk_fc_2/kernel'].zero_()
  vs.begin_call()
  v = O.sk_conv2d(ctx, x, 16, 1, True)
  vs.begin_call()
  f = O.batch_norm(ctx, O.conv2d_fixed_padding(ctx, x, 32, 3, 1), True, relu=True)
  assert torch.allclose(v, (f[:, :16] + f[:, 16:]) / 2, atol=1e-6)


def test_mixup_rules():
  x = torch.arange(4 * 3.).view(4, 1, 1, 3)
  y = F.one_hot(torch.tensor([0, 1, 2, 3]), 4).float()
  one = torch.ones(2)
  mx, my, _ = O.mixup(x, y, one, keep_batch_size=False)
  assert torch.equal(mx, x[:2]) and torch.equal(my, y[:2])               # lambda = 1 is the identity
  lam = _t([0.25, 0.75])
  mx, my, _ = O.mixup(x, y, lam, keep_batch_size=False)
  assert torch.allclose(mx[0], 0.25 * x[0] + 0.75 * x[2]) and torch.allclose(my[1], 0.75 * y[1] + 0.25 * y[3])
  lam2 = _t([0.5, 0.0])
  t = torch.rand(4, 4)
  mx, my, mt = O.mixup(x, y, lam, keep_batch_size=True, y_t=t, lam2=lam2)
  assert mx.shape[0] == 4
  assert torch.allclose(mx[2], 0.5 * x[0] + 0.5 * x[3])                   # second half pairs x1 with reverse(x2)
  assert torch.allclose(mx[3], x[2])
  assert torch.allclose(mt[2], 0.5 * y[0] + 0.5 * t[3])                   # the reference's y1 (not y1_t) quirk, :154


def test_losses():
  logits = _t([[2.0, 1.0, 0.0], [0.0, 0.0, 0.0]])
  onehot = _t([[1, 0, 0], [0, 0, 1]])
  lp = torch.log_softmax(logits, 1)
  assert abs(float(O.softmax_cross_entropy(logits, onehot)) - float(-(lp[0, 0] + lp[1, 2]) / 2)) < 1e-6
  eps = 0.1
  tgt = onehot * (1 - eps) + eps / 3
  assert abs(float(O.softmax_cross_entropy(logits, onehot, eps)) - float(-(tgt * lp).sum(1).mean())) < 1e-6
  teacher = torch.softmax(_t([[1.0, 2.0, 3.0], [3.0, 2.0, 1.0]]) / 2.0, 1)
  kd = O.kd_loss(logits, teacher, 2.0)
  exp = 4.0 * float(-(teacher * torch.log_softmax(logits / 2.0, 1)).sum(1).mean())
  assert abs(float(kd) - exp) < 1e-6
  oh, te = O.split_kd_labels(torch.cat([onehot, _t([[1.0, 2.0, 3.0], [3.0, 2.0, 1.0]])], 1), 2.0)
  assert torch.equal(oh, onehot) and torch.allclose(te, teacher)
  sig = O.get_sup_loss(logits, onehot, 'sigmoid')
  assert abs(float(sig) - float(F.binary_cross_entropy_with_logits(logits, onehot, reduction='sum') / 2)) < 1e-6


def test_l2_set_and_momentum():
  m = O.Model(50, num_classes=1001)
  m(torch.zeros(1, 64, 64, 3), False)
  tv = m.trainable_variables()
  manual = sum(0.5 * float((v.double() ** 2).sum()) for n, v in tv.items() if 'batch_normalization' not in n)
  assert abs(float(O.l2_loss(tv, 1e-4)) - 1e-4 * manual) < 1e-6 * manual * 1e-4 + 1e-9
  w, a = [_t([1.0, 2.0])], [_t([0.5, -0.5])]
  O.momentum_step(w, [_t([0.1, 0.2])], a, lr=0.1, momentum=0.9)
  assert torch.allclose(a[0], _t([0.55, -0.25])) and torch.allclose(w[0], _t([1 - 0.055, 2 + 0.025]))


def test_learning_rate_schedules():
  n, b = 1281167, 1024
  cos = O.learning_rate_with_decay('cosine', b, b, n, 2.0, 0.94, 1e-4, [30, 60, 80, 90], [1, .1, .01, .001, 1e-4], 0.4,
                                   warmup_epochs=5, train_epochs=120)
  bpe = n / b
  ws = int(bpe * 5)
  assert cos(0) == 0.0 and abs(cos(ws // 2) - 0.4 * (ws // 2) / ws) < 1e-9
  assert abs(cos(ws) - 0.4) < 1e-9                                     # batch_denom == batch -> no extra scaling
  total = int(bpe * 120) - ws
  assert abs(cos(ws + total // 2) - 0.2) < 1e-3 and cos(ws + total + 10) < 1e-12
  ex = O.learning_rate_with_decay('exponential', 32, 32, n, 2.0, 0.94, 1e-4, [], [], 0.01)
  ds = int(n / 32 * 2.0)
  assert ex(0) == 0.01 and ex(ds - 1) == 0.01 and abs(ex(ds) - 0.0094) < 1e-12
  pw = O.learning_rate_with_decay('piecewise', b, b, n, 2.0, 0.94, 1e-4, [30, 60], [1, 0.1, 0.01], 0.4)
  assert pw(int(bpe * 30)) == 0.4 and abs(pw(int(bpe * 30) + 1) - 0.04) < 1e-12 and abs(pw(10 ** 7) - 0.004) < 1e-12
  po = O.learning_rate_with_decay('polynomial', b, b, n, 2.0, 0.94, 1e-4, [], [], 0.4)
  assert abs(po(int(bpe * 2.0)) - 1e-4) < 1e-12 and abs(po(10 ** 7) - 1e-4) < 1e-12
  kp = O.keep_prob_decay(1.0, 0.9, 100)
  assert kp(0) == 1.0 and abs(kp(50) - 0.95) < 1e-12 and abs(kp(500) - 0.9) < 1e-12


def test_dropblock_mask_is_shared_and_renormalised():
  x = torch.ones(2, 3, 9, 9)
  u = torch.ones(1, 3, 3, 3)
  u[0, :, 1, 1] = 0.0                                                  # one seed per channel, centre
  y = O.dropblock(x, 0.9, 7, 1.0, True, u)
  zero = (y == 0)
  assert bool(zero[0].eq(zero[1]).all())                               # same mask for every image (:224,:228)
  assert int(zero[0, 0].sum()) == 49                                   # 7x7 block
  kept = 81 - 49
  assert abs(float(y.max()) - 81 / kept) < 1e-5                        # global renormalisation (:245-250)
  assert O.dropblock(x, 1.0, 7, 1.0, True, u) is x and O.dropblock(x, 0.5, 7, 1.0, False, u) is x
