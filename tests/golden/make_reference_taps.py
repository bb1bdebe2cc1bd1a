#!/usr/bin/env python
"""Generate tests/golden/reference_taps.json by RUNNING THE REFERENCE'S OWN NETWORK CODE, unmodified, under the torch-
backed `tensorflow` stand-in (oracle/tf_shim).  Run in the build container (needs /root/reference); the GPU box and the
test-suite only read the committed fixture.

What runs from /root/reference (imported as-is, nothing copied):
  functions/model_fns.py   Model (ImageNet defaults), get_block_sizes, learning_rate_with_decay, keep_prob_decay
  nets/resnet_model.py     Model.__call__, block_layer, _bottleneck_block_v1
  nets/blocks.py           sk_conv2d, se_block, anti_aliased_downsample, dropblock, generalized_mean_pooling
  nets/model_helper.py     conv2d_fixed_padding, fixed_padding, batch_norm
  losses/cls_losses.py     get_sup_loss
  utils/data_util.py       mixup
Every variable (and moving statistic) is overwritten with tests/golden/name_seeded.value_for(name, shape) after the
first (variable-creating) call; the recorded forward passes are second calls with reuse=True.

usage: python tests/golden/make_reference_taps.py [--check]     (--check: regenerate and compare with the file)
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
OUT = os.path.join(HERE, 'reference_taps.json')

sys.path.insert(0, HERE)
from name_seeded import tap_summary, value_for  # noqa: E402

CONFIGS = {
    # name: (Model kwargs, use_resnet_d, input size, extra call kwargs)
    'r50v1': (dict(resnet_size=50), False, 64),
    'r50v1-d': (dict(resnet_size=50), True, 64),
    'a-r50': (dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                   anti_alias_filter_size=3), False, 64),
    'a-r50-d': (dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                     anti_alias_filter_size=3), True, 64),
    'a-r152': (dict(resnet_size=152, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                    anti_alias_filter_size=3, bl_alpha=1, bl_beta=2), False, 64),
    'se-proj': (dict(resnet_size=50, use_se_block=True, anti_alias_type='proj', anti_alias_filter_size=3), False, 64),
    'r101v1-gem-emb': (dict(resnet_size=101, pool_type='gem', embedding_size=128, zero_gamma=True), False, 64),
    'r50v1-nodown-flatten-sigmoid': (dict(resnet_size=50, no_downsample=True, pool_type='flatten', loss_type='sigmoid'),
                                     False, 32),
}
BATCH = 2


def install_import_hooks():
  """`tensorflow` -> oracle/tf_shim; its submodules, absl, tensorflow_hub, hyperdash -> inert stand-ins"""
  sys.path.insert(0, os.path.join(ROOT, 'oracle', 'tf_shim'))
  sys.path.insert(0, REF)
  fake = mock.MagicMock(name='inert')
  fake.__path__ = []

  class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
      return fake

    def exec_module(self, module):
      pass

  class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
      top = name.split('.')[0]
      if top == 'tensorflow' and '.' in name:
        # the few tensorflow.python.* modules metric/ece_metric.py needs exist in the shim as real files
        rel = os.path.join(ROOT, 'oracle', 'tf_shim', *name.split('.'))
        if os.path.isdir(rel) or os.path.exists(rel + '.py'):
          return None
      if top in ('absl', 'tensorflow_hub', 'hyperdash') or (top == 'tensorflow' and '.' in name):
        return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
      return None
  sys.meta_path.insert(0, _Finder())


def seeded_input(size, seed=1):
  rng = np.random.default_rng(seed)
  img = rng.integers(0, 256, size=(BATCH, size, size, 3)).astype(np.float64)
  return img - np.array([123.68, 116.78, 103.94])


def overwrite_variables(tf):
  for name, v in tf.shim_state().variables.items():
    v.assign(torch.from_numpy(value_for(name, list(v.t.shape))))


def run_config(tf, model_fns, name):
  kw, use_d, size = CONFIGS[name]
  tf.reset_default_graph()
  x = tf.constant(seeded_input(size), tf.float32)
  model = model_fns.Model(num_classes=1001, **kw)
  model(x, True, use_resnet_d=use_d)                      # creates the variables (reference initialisers)
  st = tf.shim_state()
  out = dict(variables=[[n, list(v.t.shape), bool(v.trainable)] for n, v in st.variables.items()],
             data_format=model.data_format, block_sizes=list(model.block_sizes), block_strides=list(model.block_strides),
             input_size=size, batch=BATCH)
  # initial values the reference's initialisers gave: gammas that start at zero (zero_gamma), dense bias
  out['zero_gammas'] = [n for n, v in st.variables.items() if n.endswith('gamma') and float(v.t.abs().sum()) == 0.0]
  bias = [v for n, v in st.variables.items() if n.endswith('dense/bias')]
  out['dense_bias_init'] = float(bias[0].t[0]) if bias else None
  overwrite_variables(tf)
  for mode, training in (('train', True), ('eval', False)):
    st.named.clear()
    st.update_ops = []
    logits = model(x, training, reuse=True, use_resnet_d=use_d)
    rec = dict(taps={k: tap_summary(v.numpy()) for k, v in st.named.items()}, logits=logits.numpy().tolist())
    if training:   # the UPDATE_OPS the train op would run: new moving statistics of three BN layers
      ups = {}
      for var, val in st.update_ops:
        ups[var.name] = val
      keys = sorted(ups)
      for k in (keys[0], keys[len(keys) // 2], keys[-1]):
        rec.setdefault('moving_updates', {})[k] = tap_summary(ups[k].numpy(), 16)
      rec['n_update_ops'] = len(st.update_ops)
    out[mode] = rec
  if name == 'r101v1-gem-emb':
    st.named.clear()
    emb = model(x, False, reuse=True, use_resnet_d=use_d, return_embedding=True)
    out['embedding'] = emb.numpy().tolist()
  return out


def run_dropblock(tf, model_fns):
  """ResNet-50 v1 at 224 x 224 (stage 4 is 7 x 7, the smallest map DropBlock's 7 x 7 block fits), keep_prob 0.9"""
  tf.reset_default_graph()
  x = tf.constant(seeded_input(224, seed=3), tf.float32)
  model = model_fns.Model(resnet_size=50, num_classes=1001)
  model(x, True, keep_prob=0.9)
  overwrite_variables(tf)
  st = tf.shim_state()
  st.named.clear()
  st.uniform_draws = []
  st.rng = np.random.default_rng(77)
  logits = model(x, True, reuse=True, keep_prob=0.9)
  return dict(input_size=224, batch=BATCH, keep_prob=0.9, rng_seed=77,
              draw_shapes=[list(d.shape) for d in st.uniform_draws],
              taps={k: tap_summary(v.numpy()) for k, v in st.named.items()}, logits=logits.numpy().tolist())


def run_losses(tf):
  """get_sup_loss, mixup (types 1 and 2, with the teacher quirk of utils/data_util.py:154), LR / keep-prob schedules"""
  from losses import cls_losses
  from utils import data_util
  from functions import model_fns
  rng = np.random.default_rng(11)
  out = {}
  logits = rng.normal(0, 2, size=(6, 1001))
  onehot = np.eye(1001)[rng.integers(1, 1001, size=6)]
  soft = rng.dirichlet(np.ones(1001), size=6)
  out['loss_inputs_seed'] = 11
  for ls in (0.0, 0.1):
    p = dict(cls_loss_type='softmax', label_smoothing=ls)
    out['softmax_ce_ls%g' % ls] = float(cls_losses.get_sup_loss(tf.constant(logits, tf.float32), tf.constant(onehot, tf.float32),
                                                               None, 1001, p))
    out['softmax_ce_soft_ls%g' % ls] = float(cls_losses.get_sup_loss(tf.constant(logits, tf.float32),
                                                                    tf.constant(soft, tf.float32), None, 1001, p))
  out['sigmoid_ce'] = float(cls_losses.get_sup_loss(tf.constant(logits, tf.float32), tf.constant(onehot, tf.float32), None,
                                                    1001, dict(cls_loss_type='sigmoid', label_smoothing=0.0)))
  # mixup
  x = rng.normal(0, 50, size=(8, 6, 6, 3))
  y = np.eye(11)[rng.integers(0, 11, size=8)]
  yt = rng.dirichlet(np.ones(11), size=8)
  for keep, tag in ((False, 'type1'), (True, 'type2')):
    tf.reset_default_graph()
    tf.shim_state().rng = np.random.default_rng(4)
    mx, my, myt = data_util.mixup(tf.constant(x, tf.float32), tf.constant(y, tf.float32), alpha=0.2,
                                  keep_batch_size=keep, y_t=tf.constant(yt, tf.float32))
    draws = [d.numpy().tolist() for d in tf.shim_state().beta_draws]
    out['mixup_' + tag] = dict(lams=draws, x=tap_summary(mx.numpy(), 32), y=my.numpy().tolist(), y_t=myt.numpy().tolist())
  out['mixup_inputs_seed'] = 11
  # learning-rate schedules (functions/model_fns.py:36-95) on the recipe of scripts/train_assemble_from_scratch.sh
  sched = {}
  steps = [0, 1, 100, 6254, 6255, 6256, 40000, 150000, 749999, 750600, 900000]
  common = dict(batch_size=1024, batch_denom=1024, num_images=1281167, num_epochs_per_decay=2.0,
                learning_rate_decay_factor=0.94, end_learning_rate=1e-4, piecewise_lr_boundary_epochs=[30, 60, 80, 90],
                piecewise_lr_decay_rates=[1, 0.1, 0.01, 0.001, 1e-4], base_lr=0.4)
  for kind in ('exponential', 'fixed', 'polynomial', 'piecewise', 'cosine'):
    for warm in (0, 5):
      fn = model_fns.learning_rate_with_decay(kind, warmup_epochs=warm, train_epochs=600, **common)
      sched['%s_warm%d' % (kind, warm)] = [float(fn(tf.constant(s))) for s in steps]
  out['lr'] = dict(steps=steps, values=sched)
  kp = model_fns.keep_prob_decay(1.0, 0.9, 750600)
  out['keep_prob'] = [float(kp(tf.constant(s))) for s in steps]
  return out


def generate():
  install_import_hooks()
  import tensorflow as tf
  assert 'tf_shim' in tf.__file__, tf.__file__
  from functions import model_fns
  out = {'_generator': 'tests/golden/make_reference_taps.py', '_reference': REF, 'models': {}}
  for name in CONFIGS:
    out['models'][name] = run_config(tf, model_fns, name)
    print(name, len(out['models'][name]['variables']), 'variables,', len(out['models'][name]['train']['taps']), 'taps')
  out['dropblock_r50v1'] = run_dropblock(tf, model_fns)
  out['losses'] = run_losses(tf)
  return out


def main():
  out = generate()
  if '--check' in sys.argv:
    old = json.load(open(OUT))
    a, b = json.dumps(out, sort_keys=True), json.dumps(old, sort_keys=True)
    assert a == b, 'regenerated fixture differs from the committed file'
    print('fixture reproduces')
    return
  json.dump(out, open(OUT, 'w'), sort_keys=True)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
