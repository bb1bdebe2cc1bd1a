#!/usr/bin/env python
"""Generate tests/golden/reference_step.json by RUNNING MORE OF THE REFERENCE, unmodified, under the torch-backed
`tensorflow` stand-in (oracle/tf_shim) -- the glue round 2 had only restated:

  nets/run_loop_classification.py:60-234   resnet_model_fn: KD label split, mixup call, loss assembly (CE + L2 + KD),
                                           learning-rate / keep-prob wiring, train metrics (accuracy, top-5, ECE)
  nets/optimizer_setting.py:23-38          get_train_op: loss scaling, MomentumOptimizer, UPDATE_OPS grouping
  metric/ece_metric.py:171-298             ece(): streaming 10-bin expected calibration error
  preprocessing/imagenet_preprocessing.py  preprocess_image (evaluation and training branches), central_crop,
                                           _aspect_preserving_resize / _smallest_size_at_least, mean_image_subtraction

Run in the build container (needs /root/reference); tests and the GPU box only read the committed fixture.
usage: python tests/golden/make_reference_step.py [--check]
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_taps as T  # noqa: E402  (import hooks, variable overwrite, seeded inputs)
from name_seeded import tap_summary  # noqa: E402

OUT = os.path.join(HERE, 'reference_step.json')

ASSEMBLE = dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3)
# name: (Model kwargs, use_resnet_d, input size, images fed, hyper-parameters of the step)
STEP_CONFIGS = {
    # BASELINE config 2 style: ResNet-50, label smoothing, weight decay; loss_scale 128 must change nothing but rounding
    'r50v1-ls': (dict(resnet_size=50), False, 64, 4,
                 dict(label_smoothing=0.1, weight_decay=1e-4, momentum=0.9, base_lr=0.05, loss_scale=1, kd_temp=0,
                      mixup_type=0)),
    'r50v1-ls-scale128': (dict(resnet_size=50), False, 64, 4,
                          dict(label_smoothing=0.1, weight_decay=1e-4, momentum=0.9, base_lr=0.05, loss_scale=128,
                               kd_temp=0, mixup_type=0)),
    # BASELINE config 4: Assemble-ResNet-50 (+D) + mixup type 1 (2B images in, B out) + label smoothing (8 images in:
    # with 4 the network sees a batch of 2, whose batch norms over 2..8 values turn a 1e-16 rounding into 1e-4 by step 2)
    'a-r50-d-mixup1-ls': (ASSEMBLE, True, 64, 8,
                          dict(label_smoothing=0.1, weight_decay=4e-5, momentum=0.9, base_lr=0.02, loss_scale=1,
                               kd_temp=0, mixup_type=1)),
    # BASELINE config 5 glue: knowledge distillation (T = 2 so that T^2 and 1/T are visible) + mixup type 2 (the
    # utils/data_util.py:154 teacher quirk) on the small network
    'r50v1-kd-mixup2': (dict(resnet_size=50), False, 64, 4,
                        dict(label_smoothing=0.0, weight_decay=1e-4, momentum=0.9, base_lr=0.05, loss_scale=1,
                             kd_temp=2.0, mixup_type=2)),
}
STEPS = 2


def params(tf, kw, use_d, hp, batch):
  p = dict(resnet_size=50, data_format=None, resnet_version=1, zero_gamma=False, use_se_block=False, use_sk_block=False,
           no_downsample=False, anti_alias_filter_size=0, anti_alias_type='', bn_momentum=0.9, embedding_size=0,
           pool_type='gap', bl_alpha=2, bl_beta=4, dtype=tf.float32, cls_loss_type='softmax', use_resnet_d=use_d,
           kd_temp=hp['kd_temp'], mixup_type=hp['mixup_type'], label_smoothing=hp['label_smoothing'],
           weight_decay=hp['weight_decay'], momentum=hp['momentum'], loss_scale=hp['loss_scale'])
  p.update(kw)
  return p


def run_step_config(tf, model_fns, run_loop, name):
  kw, use_d, size, n_in, hp = STEP_CONFIGS[name]
  tf.reset_default_graph()
  st = tf.shim_state()
  st.track_grad = True
  rng = np.random.default_rng(21)
  x = rng.integers(0, 256, size=(n_in, size, size, 3)).astype(np.float64) - np.array([123.68, 116.78, 103.94])
  labels = rng.integers(1, 1001, size=n_in)
  teacher_logits = rng.normal(0, 3, size=(n_in, 1001))
  p = params(tf, kw, use_d, hp, n_in)
  # create the variables with the reference's own constructor path, then overwrite them with the name-seeded values
  # (so no weights are stored); the step itself then runs under reuse
  model = model_fns.Model(p['resnet_size'], p['data_format'], num_classes=1001, resnet_version=p['resnet_version'],
                          zero_gamma=p['zero_gamma'], use_se_block=p['use_se_block'], use_sk_block=p['use_sk_block'],
                          no_downsample=p['no_downsample'], anti_alias_filter_size=p['anti_alias_filter_size'],
                          anti_alias_type=p['anti_alias_type'], bn_momentum=p['bn_momentum'],
                          embedding_size=p['embedding_size'], pool_type=p['pool_type'], bl_alpha=p['bl_alpha'],
                          bl_beta=p['bl_beta'], dtype=p['dtype'], loss_type=p['cls_loss_type'])
  model(tf.constant(x, tf.float32), True, use_resnet_d=use_d)
  T.overwrite_variables(tf)
  st.update_ops = []
  st.named.clear()
  st.reuse = True
  st.rng = np.random.default_rng(4)      # the Beta(0.2, 0.2) draws of mixup
  lr_fn = model_fns.learning_rate_with_decay('cosine', batch_size=1024, batch_denom=1024, num_images=1281167,
                                             num_epochs_per_decay=2.0, learning_rate_decay_factor=0.94,
                                             end_learning_rate=1e-4, piecewise_lr_boundary_epochs=[30, 60, 80, 90],
                                             piecewise_lr_decay_rates=[1, 0.1, 0.01, 0.001, 1e-4], base_lr=hp['base_lr'],
                                             warmup_epochs=0, train_epochs=120)
  out = dict(input_size=size, images_in=n_in, seed=21, beta_seed=4, hp=hp, bn_momentum=p['bn_momentum'], steps=[],
             labels=labels.tolist())
  for step in range(STEPS):
    st.named.clear()
    st.beta_draws = []
    if hp['kd_temp'] > 0:
      lab = tf.constant(np.concatenate([np.eye(1001)[labels], teacher_logits], 1), tf.float32)
    else:
      lab = tf.constant(labels, tf.int32)
    spec = run_loop.resnet_model_fn({'image': tf.constant(x, tf.float32)}, lab, 1001, tf.estimator.ModeKeys.TRAIN,
                                    model_fns.Model, lr_fn, None, None, p)
    rec = dict(loss=float(spec.loss), cross_entropy=float(st.named['cross_entropy']),
               learning_rate=float(st.named['learning_rate']), global_step_after=int(st.variables['global_step'].t),
               lams=[d.numpy().tolist() for d in st.beta_draws])
    if hp['kd_temp'] > 0:
      rec['cross_entropy_kd'] = float(st.named['cross_entropy_kd'])
    if hp['mixup_type'] == 0:
      rec['metrics'] = {k: float(v[1]) for k, v in spec.eval_metric_ops.items()}
      rec['classes'] = spec.predictions['classes'].numpy().tolist()
    else:
      assert spec.eval_metric_ops is None
    # the loss-scale variant must equal its twin: a thinned record is enough (fixture size)
    keep = (lambda i: i % 13 == 0) if hp['loss_scale'] != 1 else (lambda i: True)
    rec['grads'] = {n: tap_summary(g.numpy(), 6) for i, (n, g) in enumerate(st.last_grads.items()) if keep(i)}
    if step == STEPS - 1:    # after the last step: every variable, moving statistic and Momentum slot
      rec['variables_after'] = {n: tap_summary(v.t.detach().numpy(), 6) for i, (n, v) in enumerate(st.variables.items())
                                if n != 'global_step' and keep(i)}
      rec['momentum_after'] = {n: tap_summary(a.numpy(), 4) for i, (n, a) in enumerate(st.opt_slots.items()) if keep(i)}
    assert not st.update_ops, 'tf.group left UPDATE_OPS pending'
    out['steps'].append(rec)
  out['n_grads'] = len(st.last_grads)
  return out


# ---- ECE -------------------------------------------------------------------------------------------------------
def run_ece(tf):
  from metric import ece_metric
  rng = np.random.default_rng(31)
  tf.reset_default_graph()
  st = tf.shim_state()
  out = dict(seed=31, batches=[])
  edge = np.array([0.0, 0.1, 0.2, 0.3, 0.5, 0.9, 1.0], dtype=np.float32)      # confidences exactly ON bin edges
  for b in range(3):
    n = 40 + 13 * b
    conf = rng.uniform(0.0, 1.0, size=n).astype(np.float32)
    conf[:len(edge)] = edge
    pred = rng.integers(0, 5, size=n)
    label = np.where(rng.random(n) < 0.6, pred, rng.integers(0, 5, size=n))
    st.scope_counts = {}          # re-open the SAME 'ece' scope: one metric, three session.run calls
    value, update = ece_metric.ece(tf.constant(conf, tf.float32), tf.constant(pred, tf.int64), tf.constant(label, tf.int64))
    acc = {k.rsplit('/', 1)[1]: v.t.numpy().tolist() for k, v in st.metric_vars.items()}
    out['batches'].append(dict(conf=[float(c) for c in conf], pred=pred.tolist(), label=label.tolist(),
                               ece_update=float(update), ece_value=float(value), accumulators=acc))
  return out


# ---- preprocessing ---------------------------------------------------------------------------------------------
def run_preprocessing(tf):
  from preprocessing import imagenet_preprocessing as P
  tf.set_compute_dtype(torch.float32)     # the graph's arithmetic type: a float32 rounding decides resize targets
  try:
    tf.reset_default_graph()
    st = tf.shim_state()
    out = dict(seed=41, eval=[], train=[], sizes=[])
    # _smallest_size_at_least on shapes whose float32 product lands on / next to an integer
    for (h, w, m) in [(500, 375, 256), (375, 500, 256), (333, 500, 292), (256, 256, 256), (375, 375, 256), (300, 300, 257),
                      (224, 224, 256), (97, 131, 73), (480, 640, 366), (1, 1, 256), (231, 640, 256), (1200, 900, 292)]:
      nh, nw = P._smallest_size_at_least(tf.constant(h), tf.constant(w), m)
      out['sizes'].append([h, w, m, int(nh), int(nw)])
    rng = np.random.default_rng(41)
    shapes = [(75, 100), (100, 67), (64, 64), (58, 160), (49, 49), (1, 1)]
    for k, (h, w) in enumerate(shapes):
      img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
      for (side, crop_type) in ((48, 0), (40, 1)):
        res, _ = P.preprocess_image(tf.constant(img), None, side, side, 3, is_training=False, crop_type=crop_type)
        out['eval'].append(dict(image=k, h=h, w=w, side=side, crop_type=crop_type, out=tap_summary(res.numpy(), 96)))
    # training branch: the box / flip draws are the shim's own (recorded); everything after them is the reference's
    box_rng = np.random.default_rng(43)

    def sampler(H, W, min_cov):
      if min_cov >= 1.0:
        return 0, 0, H, W
      hh = int(box_rng.integers(max(1, H // 3), H + 1))
      ww = int(box_rng.integers(max(1, W // 3), W + 1))
      return int(box_rng.integers(0, H - hh + 1)), int(box_rng.integers(0, W - ww + 1)), hh, ww
    st.box_sampler = sampler
    st.rng = np.random.default_rng(44)
    for k, (h, w) in enumerate(shapes[:5]):
      img = np.random.default_rng(100 + k).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
      for use_random_crop in (True, False):
        st.window_draws = []
        res, _ = P.preprocess_image(tf.constant(img), None, 32, 32, 3, is_training=True, use_random_crop=use_random_crop)
        box = [d for d in st.window_draws if d[0] == 'box'][0]
        flip = [d for d in st.window_draws if d[0] == 'flip'][0][1]
        out['train'].append(dict(image_seed=100 + k, h=h, w=w, side=32, use_random_crop=use_random_crop,
                                 window=[box[1], box[2], box[3], box[4], flip], out=tap_summary(res.numpy(), 96)))
    # central_crop + mean_image_subtraction on their own, and the error the reference raises
    a = rng.normal(0, 30, size=(11, 14, 3)).astype(np.float32)
    out['central_crop'] = dict(inp=tap_summary(a, 8), out=P.central_crop(tf.constant(a, tf.float32), 6, 9).numpy().tolist())
    try:
      P.mean_image_subtraction(tf.constant(a[None], tf.float32), P.CHANNEL_MEANS, 3)
      out['mean_sub_rank_error'] = None
    except ValueError as e:
      out['mean_sub_rank_error'] = str(e)
    out['channel_means'] = list(P.CHANNEL_MEANS)
    return out
  finally:
    tf.set_compute_dtype(torch.float64)


def generate(only=None):
  T.install_import_hooks()
  import tensorflow as tf
  assert 'tf_shim' in tf.__file__, tf.__file__
  from functions import model_fns
  from nets import run_loop_classification as run_loop
  out = {'_generator': 'tests/golden/make_reference_step.py', '_reference': T.REF, 'steps': {}}
  for name in STEP_CONFIGS:
    if only is not None and name not in only:
      continue
    out['steps'][name] = run_step_config(tf, model_fns, run_loop, name)
    s0 = out['steps'][name]['steps'][0]
    print(name, 'loss %.6f' % s0['loss'], 'ce %.6f' % s0['cross_entropy'], out['steps'][name]['n_grads'], 'gradients')
  out['ece'] = run_ece(tf)
  print('ece', [b['ece_update'] for b in out['ece']['batches']])
  out['preprocessing'] = run_preprocessing(tf)
  print('preprocessing', len(out['preprocessing']['eval']), 'eval,', len(out['preprocessing']['train']), 'train')
  return out


def main():
  only = None
  if '--only' in sys.argv:       # --check --only r50v1-ls,...: regenerate and compare a subset of the step configurations
    only = sys.argv[sys.argv.index('--only') + 1].split(',')
  out = generate(only)
  if '--check' in sys.argv:
    old = json.load(open(OUT))
    if only is not None:
      old['steps'] = {k: v for k, v in old['steps'].items() if k in only}
    a, b = json.dumps(out, sort_keys=True), json.dumps(old, sort_keys=True)
    assert a == b, 'regenerated fixture differs from the committed file'
    print('fixture reproduces')
    return
  json.dump(out, open(OUT, 'w'), sort_keys=True)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
