"""Host logic of the widened rows (SURVEY 8f + the remaining loss / pooling variants) through the CPU double:
DropBlock, GeM / flatten pooling, sigmoid loss, on-device eval metrics, TF-layout checkpoint import/export."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import model_parity as mp
from tests import util


def _db_provider(seed, log):
  g = torch.Generator().manual_seed(seed)

  def provide(shape):
    u = torch.rand(shape, generator=g)
    log.append(u)
    return u
  return provide


@pytest.mark.parametrize('name', ['a-r50', 'se-proj'])
def test_dropblock_forward_matches_oracle(cpu_double, name):
  from oracle import assembled_oracle as O
  om, pm = mp.make_pair(name, 'cpu', 2, 224)      # DropBlock needs >= 7x7 maps in stage 4 -> 224x224 input
  _, x, _ = mp.inputs(2, 224)
  log = []
  lo = om(x, True, keep_prob=0.8, dropblock_uniforms=_db_provider(5, log)).detach()
  assert len(log) >= 10, 'dropblock must fire in stages 3 and 4'
  uni = [u[0].permute(1, 2, 0).contiguous() for u in log]          # [1,C,h,w] -> [h,w,C]
  lp = pm(x, True, keep_prob=0.8, dropblock_uniforms=uni).float()
  e = util.rel_l2(lp, lo)
  assert e <= 8e-2, 'logits rel_l2 %.3e' % e
  # keep_prob == 1 and eval mode are the identity (nets/blocks.py:205-213)
  l1 = pm(x, False, keep_prob=0.8).float()
  l2 = pm(x, False).float()
  assert torch.equal(l1, l2)
  with pytest.raises(ValueError):
    pm(x, True, keep_prob=0.0)


def test_dropblock_train_step_and_schedule(cpu_double):
  from assembled_cnn_amd import train
  hp = train.HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                     zero_gamma=True, use_dropblock=True, dropblock_kp=[0.9, 0.8], train_epochs=1, batch_size=4,
                     num_images_train=40, learning_rate_decay_type='fixed', base_learning_rate=0.001)
  tr = train.Trainer(hp, device='cpu')
  kp = tr.keep_prob_fn
  assert kp(0) == 0.9 and abs(kp(5) - 0.85) < 1e-9 and abs(kp(10) - 0.8) < 1e-9 and abs(kp(99) - 0.8) < 1e-9
  img, _, labels = mp.inputs(4, 224)
  l0 = float(tr.train_step(img, labels).mean())
  assert tr.last['keep_prob'] == 0.9 and np.isfinite(l0)
  tr.train_step(img, labels)
  assert abs(tr.last['keep_prob'] - 0.89) < 1e-9
  assert bool(torch.isfinite(tr.model.arena.w32).all()) and bool(torch.isfinite(tr.model.arena.g32).all())


@pytest.mark.parametrize('pool', ['gem', 'flatten'])
def test_pool_types(cpu_double, pool):
  from assembled_cnn_amd.model import Model
  from oracle import assembled_oracle as O
  om = O.Model(50, num_classes=37, emulate_bf16=True, zero_gamma=True, pool_type=pool)
  pm = Model(50, num_classes=37, device='cpu', zero_gamma=True, pool_type=pool)
  x = mp.inputs(4, 64)[1]
  om(x[:2], True)
  om.vars.pending_updates = {}
  pm.build((64, 64))
  util.load_oracle_into_product(om, pm)
  lo = om(x, True)
  lp = pm(x, True)
  assert util.rel_l2(lp.float(), lo.detach()) <= 6e-2
  # gradient flows through the pooling variant
  from assembled_cnn_amd import ops
  oh = ops.onehot(torch.tensor([1, 2, 3, 4], dtype=torch.int32), 4, 37)
  rows, dz = ops.softmax_ce(pm.logits_padded, pm.ldc, oh, None, 4, 37, 0.0, 0.0, 1.0, pm.ldc)
  pm.backward(dz)
  loss = O.softmax_cross_entropy(lo, F.one_hot(torch.tensor([1, 2, 3, 4]), 37).float())
  og = torch.autograd.grad(loss, list(om.vars.trainable.values()))
  first = list(om.vars.trainable)[0]
  pg = util.product_to_oracle_grad(first, pm.arena.g(first), om.vars.trainable[first]).double().reshape(-1)
  gg = og[0].double().reshape(-1)
  assert float((pg * gg).sum() / (pg.norm() * gg.norm())) > 0.8


def test_sigmoid_loss_step(cpu_double):
  from assembled_cnn_amd import train
  from oracle import assembled_oracle as O
  hp = train.HParams(zero_gamma=True, cls_loss_type='sigmoid', learning_rate_decay_type='fixed',
                     base_learning_rate=0.001, batch_size=4)
  tr = train.Trainer(hp, device='cpu')
  om = O.Model(50, num_classes=1001, emulate_bf16=True, zero_gamma=True, loss_type='sigmoid')
  img, x, labels = mp.inputs(4, 64)
  om(x[:2], True)
  om.vars.pending_updates = {}
  tr.model.build((64, 64))
  util.load_oracle_into_product(om, tr.model)
  assert abs(float(tr.model.arena.w('resnet_model/dense/bias')[0]) + np.log(1000)) < 1e-5   # -log(C-1) bias init
  lo = om(x, True)
  ref = float(O.get_sup_loss(lo, F.one_hot(labels.long(), 1001).float(), 'sigmoid'))
  got = float(tr.train_step(img, labels)[0])
  assert abs(got - ref) <= 2e-2 * abs(ref), (got, ref)


def test_eval_metrics_on_device(cpu_double):
  from assembled_cnn_amd import train
  hp = train.HParams(zero_gamma=True, batch_size=8)
  tr = train.Trainer(hp, device='cpu')
  _, x, labels = mp.inputs(8, 64)
  tr.eval_reset()
  pred = tr.eval_step(x, labels)
  logits = tr.model.logits_padded.view(8, -1)[:, :1001]
  assert torch.equal(pred.long(), logits.argmax(1))
  # make a second batch whose labels are the predictions -> accuracy (8 + hits) / 16
  tr.eval_step(x, pred)
  r = tr.eval_result()
  hits = float((logits.argmax(1) == labels.long()).sum())
  assert abs(r['accuracy'] - (8 + hits) / 16) < 1e-6 and r['count'] == 16
  assert r['accuracy_top_5'] >= r['accuracy'] and 0.0 <= r['ece'] <= 1.0
  # ECE against a direct evaluation of metric/ece_metric.py's formula
  conf = torch.softmax(logits, 1).max(1).values
  conf = torch.cat([conf, conf])
  correct = torch.cat([(logits.argmax(1) == labels.long()).float(), torch.ones(8)])
  ece = 0.0
  for b in range(10):
    lo_, hi_ = (-1e-7 if b == 0 else b / 10), (1 + 1e-7 if b == 9 else (b + 1) / 10)
    sel = (conf > lo_) & (conf <= hi_)
    if sel.any():
      ece += float(sel.sum()) / 16 * abs(float(correct[sel].mean()) - float(conf[sel].mean()))
  assert abs(r['ece'] - ece) < 1e-5


def test_checkpoint_roundtrip_tf_layouts(cpu_double, tmp_path):
  from assembled_cnn_amd import checkpoint as ck
  from assembled_cnn_amd.model import Model
  from oracle import assembled_oracle as O
  kw = dict(resnet_version=2, use_sk_block=True, use_se_block=True, anti_alias_type='sconv', anti_alias_filter_size=3)
  om = O.Model(50, num_classes=1001, **kw)
  om(torch.zeros(1, 64, 64, 3), False)
  util.perturb_bn_state(om, 3)
  a = Model(50, num_classes=1001, device='cpu', **kw)
  a.build((64, 64))
  # import the oracle's variables (TF layouts: HWIO, dense [in, out]) by name
  variables = {n: t.detach().numpy() for n, t in om.vars.trainable.items()}
  variables.update({n: t.detach().numpy() for n, t in om.vars.state.items()})
  rep = ck.import_variables(a, variables)
  assert not rep['missing'] and len(rep['loaded']) == len(variables)
  a.arena.m32.copy_(torch.arange(a.arena.m32.numel(), dtype=torch.float32) % 97 - 48)     # non-trivial momentum slots
  exp = ck.export_variables(a, global_step=7)
  for n, v in variables.items():
    assert exp[n].shape == v.shape and np.allclose(exp[n], v), n
    if n in om.vars.trainable:      # '<var>/Momentum' slots of the Estimator checkpoint, same TF layout as the variable
      assert exp[n + '/Momentum'].shape == v.shape
  assert int(exp['global_step']) == 7
  c = Model(50, num_classes=1001, device='cpu', seed=4, **kw)
  c.build((64, 64))
  ck.import_variables(c, exp)
  for n in a.arena.specs:      # (the arenas pad every variable to 8 elements; compare the variables, not the padding)
    assert torch.equal(c.arena.m(n), a.arena.m(n)) and torch.equal(c.arena.w(n), a.arena.w(n)), 'resume restores ' + n
  ck.import_variables(c, {k: v for k, v in exp.items() if not k.endswith('/Momentum')})     # slots are optional on import
  # npz round trip + warm start: everything but the classifier ('dense' outside se_block) is restored
  ck.save_npz(str(tmp_path / 'm.npz'), a)
  b = Model(50, num_classes=1001, device='cpu', seed=9, **kw)
  b.build((64, 64))
  before = b.arena.w('resnet_model/dense/kernel').clone()
  rep = ck.import_variables(b, ck.load_npz(str(tmp_path / 'm.npz')), warm_start=True)
  assert rep['skipped'] == ['resnet_model/dense/kernel', 'resnet_model/dense/bias']
  assert torch.equal(b.arena.w('resnet_model/dense/kernel'), before)
  some_se = [n for n in a.arena.specs if 'se_block' in n][0]
  assert torch.equal(b.arena.w(some_se), a.arena.w(some_se))
  first = list(a.arena.specs)[0]
  assert torch.equal(b.arena.w(first), a.arena.w(first)) and torch.equal(b.arena.wb(first), a.arena.wb(first))
  # embedding head: 'embedding_dense/kernel' is a tf.layers.conv2d kernel -> [1, 1, in, emb], not a 2-D dense matrix
  oe = O.Model(50, num_classes=1001, embedding_size=64)
  oe(torch.zeros(1, 64, 64, 3), False)
  e = Model(50, num_classes=1001, device='cpu', embedding_size=64)
  e.build((64, 64))
  ve = {n: t.detach().numpy() for n, t in oe.vars.trainable.items()}
  ve.update({n: t.detach().numpy() for n, t in oe.vars.state.items()})
  ck.import_variables(e, ve)
  xe = ck.export_variables(e, include_slots=False)
  assert xe['resnet_model/embedding_dense/kernel'].shape == (1, 1, 2048, 64) == ve['resnet_model/embedding_dense/kernel'].shape
  assert xe['resnet_model/dense/kernel'].shape == (64, 1001)
  for n, v in ve.items():
    assert xe[n].shape == v.shape and np.allclose(xe[n], v), n
  with pytest.raises(ValueError):
    ck.import_variables(e, dict(ve, **{'resnet_model/embedding_dense/kernel': np.zeros((2048, 64), np.float32)}))
  # not at step 0 -> the hook does nothing
  assert ck.import_variables(b, {}, warm_start=True, global_step=5)['loaded'] == []
  with pytest.raises(KeyError):
    ck.import_variables(b, {})
