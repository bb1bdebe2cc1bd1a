"""The oracle against the REFERENCE'S OWN SOURCE.

tests/golden/reference_taps.json was produced by importing /root/reference's nets/resnet_model.py, nets/blocks.py,
nets/model_helper.py, functions/model_fns.py, losses/cls_losses.py and utils/data_util.py UNMODIFIED under the torch-
backed `tensorflow` stand-in of oracle/tf_shim (tests/golden/make_reference_taps.py; float64).  These tests rebuild the
same networks with the oracle (float64 parameters, no bf16 emulation), fill the variables from the same name-seeded
function and require

  * identical variable names, creation order, shapes and trainable flags (tf.global_variables() order);
  * every named tap of the reference graph (tf.identity(..., name)) and the logits to 1e-9 relative, training and
    inference mode, for 8 configurations incl. all BASELINE ones;
  * the UPDATE_OPS moving-statistics values, the DropBlock path (shared uniform draws), get_sup_loss, mixup types 1 / 2
    (incl. the teacher quirk of utils/data_util.py:154), the learning-rate and keep-prob schedules (also for the
    product's host-side schedule functions).

This pins the oracle's WIRING to the reference's code.  What stays pinned only by stated TensorFlow rules ([TF-sem])
is the arithmetic inside each tf op, which oracle/tf_shim restates independently of the oracle.
When /root/reference is present (the build container) the fixture is also regenerated and must reproduce."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from name_seeded import tap_summary, value_for  # noqa: E402

FIX = json.load(open(os.path.join(HERE, 'golden', 'reference_taps.json')))
BATCH = 2
REL = {'eval': 1e-9, 'train': 1e-6}   # train: batch statistics over 2..8 samples amplify float64 rounding (measured <= 1e-8)


def _input(size, seed=1):
  rng = np.random.default_rng(seed)
  img = rng.integers(0, 256, size=(BATCH, size, size, 3)).astype(np.float64)
  return torch.from_numpy(img - np.array([123.68, 116.78, 103.94]))


def _close(a, b, what, rel=1e-9):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
  scale = max(float(np.abs(b).max()), 1e-30)
  err = float(np.abs(a - b).max()) / scale
  assert err <= rel, '%s: max rel err %.3e' % (what, err)


def _check_summary(arr, ref, what, rel=1e-9):
  s = tap_summary(arr, len(ref['idx']))
  assert s['shape'] == ref['shape'], '%s: shape %s vs reference %s' % (what, s['shape'], ref['shape'])
  assert s['idx'] == ref['idx']
  _close(s['vals'], ref['vals'], what + ' samples', rel)
  _close([s['abs_sum']], [ref['abs_sum']], what + ' sum of magnitudes', rel)
  assert abs(s['sum'] - ref['sum']) <= rel * max(ref['abs_sum'], 1e-30), what + ' sum'


def _build(name, kw, use_d, size):
  from oracle import assembled_oracle as O
  fx = FIX['models'][name]
  m = O.Model(num_classes=1001, param_dtype=torch.float64, **kw)
  m(torch.zeros((BATCH, size, size, 3), dtype=torch.float64), True, use_resnet_d=use_d)
  m.vars.pending_updates = {}
  return m, fx


def _fill(m):
  with torch.no_grad():
    for n, t in m.vars.trainable.items():
      t.copy_(torch.from_numpy(value_for(n, list(t.shape))))
    for n in list(m.vars.state.keys()):
      m.vars.state[n] = torch.from_numpy(value_for(n, list(m.vars.state[n].shape)))


MODEL_KW = {
    'r50v1': (dict(resnet_size=50), False),
    'r50v1-d': (dict(resnet_size=50), True),
    'a-r50': (dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                   anti_alias_filter_size=3), False),
    'a-r50-d': (dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                     anti_alias_filter_size=3), True),
    'a-r152': (dict(resnet_size=152, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                    anti_alias_filter_size=3, bl_alpha=1, bl_beta=2), False),
    'se-proj': (dict(resnet_size=50, use_se_block=True, anti_alias_type='proj', anti_alias_filter_size=3), False),
    'r101v1-gem-emb': (dict(resnet_size=101, pool_type='gem', embedding_size=128, zero_gamma=True), False),
    'r50v1-nodown-flatten-sigmoid': (dict(resnet_size=50, no_downsample=True, pool_type='flatten', loss_type='sigmoid'),
                                     False),
}


def test_fixture_covers_the_baseline_configurations():
  assert sorted(FIX['models']) == sorted(MODEL_KW)
  assert FIX['models']['r50v1']['data_format'] == 'channels_last'


@pytest.mark.parametrize('name', sorted(MODEL_KW))
def test_oracle_equals_reference_under_shim(name):
  kw, use_d = MODEL_KW[name]
  fx = FIX['models'][name]
  size = fx['input_size']
  m, _ = _build(name, kw, use_d, size)
  # ---- variables: names, creation order, shapes, trainable flags ----
  ref_train = [(n, s) for n, s, tr in fx['variables'] if tr]
  ref_state = [(n, s) for n, s, tr in fx['variables'] if not tr]
  assert [n for n, _ in ref_train] == list(m.vars.trainable.keys()), 'trainable variable names / order'
  assert [s for _, s in ref_train] == [list(t.shape) for t in m.vars.trainable.values()], 'trainable variable shapes'
  assert [n for n, _ in ref_state] == list(m.vars.state.keys()), 'moving-statistics names / order'
  assert fx['block_sizes'] == list(m.block_sizes) and fx['block_strides'] == list(m.block_strides)
  # initialisers the reference chose: zero-gamma set and dense bias
  zg = [n for n, t in m.vars.trainable.items() if n.endswith('gamma') and float(t.abs().sum()) == 0.0]
  assert zg == fx['zero_gammas']
  bias = [t for n, t in m.vars.trainable.items() if n.endswith('dense/bias')][0]
  assert abs(float(bias[0]) - fx['dense_bias_init']) <= 1e-12
  # interleaving of trainable / non-trainable creation (tf.global_variables() order)
  order = [n for n, _, _ in fx['variables']]
  pos = {n: i for i, n in enumerate(order)}
  for n in m.vars.state:                         # a BN layer's moving stats follow its gamma / beta immediately
    base = n.rsplit('/', 1)[0]
    assert pos[n] in (pos[base + '/beta'] + 1, pos[base + '/beta'] + 2)
  # ---- forward passes ----
  _fill(m)
  x = _input(size)
  for mode, training in (('train', True), ('eval', False)):
    logits = m(x, training, use_resnet_d=use_d).detach().numpy()
    taps = {k: v.detach().numpy() for k, v in m.taps_nhwc().items()}
    ref = fx[mode]
    assert sorted(ref['taps']) == sorted(taps), '%s: tap names %s vs reference %s' % (mode, sorted(taps), sorted(ref['taps']))
    for k, r in ref['taps'].items():
      _check_summary(taps[k], r, '%s/%s tap %s' % (name, mode, k), REL[mode])
    _close(logits, ref['logits'], '%s/%s logits' % (name, mode), REL[mode])
    if training:
      assert ref['n_update_ops'] == len(m.vars.pending_updates)
      for k, r in ref['moving_updates'].items():
        _check_summary(m.vars.pending_updates[k].detach().numpy(), r, '%s moving update %s' % (name, k), REL['train'])
      m.vars.pending_updates = {}
  if 'embedding' in fx:
    emb = m(x, False, use_resnet_d=use_d, return_embedding=True).detach().numpy()
    _close(emb, fx['embedding'], name + ' embedding')


def test_dropblock_path_equals_reference():
  from oracle import assembled_oracle as O
  fx = FIX['dropblock_r50v1']
  m = O.Model(50, num_classes=1001, param_dtype=torch.float64)
  x = _input(fx['input_size'], seed=3)
  m(x[:, :64, :64], True)
  m.vars.pending_updates = {}
  _fill(m)
  rng = np.random.default_rng(fx['rng_seed'])
  draws = [torch.from_numpy(rng.uniform(0, 1, size=s)).permute(0, 3, 1, 2) for s in fx['draw_shapes']]
  assert len(draws) >= 20 and fx['draw_shapes'][-1] == [1, 1, 1, 2048]
  logits = m(x, True, keep_prob=fx['keep_prob'], dropblock_uniforms=draws).detach().numpy()
  taps = {k: v.detach().numpy() for k, v in m.taps_nhwc().items()}
  for k, r in fx['taps'].items():
    _check_summary(taps[k], r, 'dropblock tap ' + k, REL['train'])
  _close(logits, fx['logits'], 'dropblock logits', REL['train'])


def test_losses_mixup_and_schedules_equal_reference():
  from oracle import assembled_oracle as O
  from assembled_cnn_amd import train as T
  fx = FIX['losses']
  rng = np.random.default_rng(fx['loss_inputs_seed'])
  logits = torch.from_numpy(rng.normal(0, 2, size=(6, 1001)))
  onehot = torch.from_numpy(np.eye(1001)[rng.integers(1, 1001, size=6)])
  soft = torch.from_numpy(rng.dirichlet(np.ones(1001), size=6))
  for ls in (0.0, 0.1):
    _close(float(O.get_sup_loss(logits, onehot, 'softmax', ls)), fx['softmax_ce_ls%g' % ls], 'softmax CE ls=%g' % ls)
    _close(float(O.get_sup_loss(logits, soft, 'softmax', ls)), fx['softmax_ce_soft_ls%g' % ls], 'soft-target CE')
  _close(float(O.get_sup_loss(logits, onehot, 'sigmoid')), fx['sigmoid_ce'], 'sigmoid CE')
  x = torch.from_numpy(rng.normal(0, 50, size=(8, 6, 6, 3)))
  y = torch.from_numpy(np.eye(11)[rng.integers(0, 11, size=8)])
  yt = torch.from_numpy(rng.dirichlet(np.ones(11), size=8))
  for keep, tag in ((False, 'type1'), (True, 'type2')):
    r = fx['mixup_' + tag]
    lam1 = torch.tensor(r['lams'][0], dtype=torch.float64)
    lam2 = torch.tensor(r['lams'][1], dtype=torch.float64) if keep else None
    assert len(r['lams']) == (2 if keep else 1)
    mx, my, myt = O.mixup(x, y, lam1, keep_batch_size=keep, y_t=yt, lam2=lam2)
    _check_summary(mx.numpy(), r['x'], 'mixup %s images' % tag)
    _close(my.numpy(), r['y'], 'mixup %s labels' % tag)
    _close(myt.numpy(), r['y_t'], 'mixup %s teacher labels' % tag)
  common = dict(batch_size=1024, batch_denom=1024, num_images=1281167, num_epochs_per_decay=2.0,
                learning_rate_decay_factor=0.94, end_learning_rate=1e-4, piecewise_lr_boundary_epochs=[30, 60, 80, 90],
                piecewise_lr_decay_rates=[1, 0.1, 0.01, 0.001, 1e-4], base_lr=0.4)
  steps = fx['lr']['steps']
  for key, vals in fx['lr']['values'].items():
    kind, warm = key.rsplit('_warm', 1)
    for mod, label in ((O, 'oracle'), (T, 'product')):
      fn = mod.learning_rate_with_decay(kind, warmup_epochs=int(warm), train_epochs=600, **common)
      got = [fn(s) for s in steps]
      assert np.allclose(got, vals, rtol=1e-6, atol=1e-12), '%s LR schedule %s: %s vs reference %s' % (label, key, got, vals)
  for mod in (O, T):
    kp = mod.keep_prob_decay(1.0, 0.9, 750600)
    assert np.allclose([kp(s) for s in steps], fx['keep_prob'], rtol=1e-9)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference tree is only present in the build container')
def test_fixture_reproduces_from_the_reference_tree():
  """Re-runs the reference under the shim for one light configuration and compares with the committed fixture."""
  code = ('import sys, json; sys.argv=["x"]; sys.path.insert(0, %r); import make_reference_taps as g; '
          'g.install_import_hooks(); import tensorflow as tf; from functions import model_fns; '
          'print(json.dumps(g.run_config(tf, model_fns, "a-r50-d"), sort_keys=True))' % os.path.join(HERE, 'golden'))
  r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  got = json.loads(r.stdout.strip().splitlines()[-1])
  assert json.dumps(got, sort_keys=True) == json.dumps(FIX['models']['a-r50-d'], sort_keys=True)


@pytest.mark.parametrize('name', ['r50v1', 'a-r50', 'a-r50-d', 'a-r152', 'se-proj', 'r101v1-gem-emb'])
def test_product_variables_equal_reference_graph(cpu_double, name):
  """The PRODUCT's variable table (names, tf.trainable_variables() order, shapes up to the HWIO -> KRSC layout change,
  moving statistics) against the variables the reference's own code created under the shim."""
  from assembled_cnn_amd.model import Model
  kw, use_d = MODEL_KW[name]
  fx = FIX['models'][name]
  pm = Model(num_classes=1001, device='cpu', **kw)
  pm.build((fx['input_size'], fx['input_size']), use_resnet_d=use_d)
  ref_train = [(n, s) for n, s, tr in fx['variables'] if tr]
  ref_state = [(n, s) for n, s, tr in fx['variables'] if not tr]
  assert [n for n, _ in ref_train] == list(pm.arena.specs.keys())
  for (n, s), sp in zip(ref_train, pm.arena.specs.values()):
    if len(s) == 4:      # HWIO -> [Cout][R][S][Cin]
      want = (s[3], s[0], s[1], s[2])
    elif len(s) == 2:    # dense [in, out] -> [out][1][1][in]
      want = (s[1], 1, 1, s[0])
    else:
      want = tuple(s)
    assert tuple(sp.shape) == want, (n, sp.shape, s)
    assert sp.decay == ('batch_normalization' not in n)     # nets/run_loop_classification.py:166-177
  assert [n for n, _ in ref_state] == list(pm.arena.state_specs.keys())
  zg = [n for n in pm.arena.specs if n.endswith('gamma') and float(pm.arena.w(n).abs().sum()) == 0.0]
  assert zg == fx['zero_gammas']
