"""Whole-model parity helpers shared by the CPU (test double) and GPU (HIP) tiers.

Why these tolerances.  Activations are stored in bf16 (8 significant bits).  Two mathematically
identical bf16 pipelines differ by isolated 1-ulp rounding flips which a randomly initialised ReLU/BN
ResNet amplifies layer by layer; worse, a gradient that is a SUM over ReLU-masked elements changes by
~sqrt(p) relative when a fraction p of the masks flips.  Measured on the oracle ALONE (its fp32 graph
vs its bf16-emulating graph, ResNet-50, batch 8, 96x96, damped zero-gamma): logits rel-L2 2.3e-2,
gradient global rel-L2 0.43, cosine 0.92.  So whole-network checks are statistical (logits rel-L2,
per-variable gradient cosine / norm ratio, loss trajectories) while the tight, bit-level-ish checks
live in the per-kernel tests (tests/test_gpu_conv.py, tests/test_gpu_ops.py) where both sides see
identical inputs.  Block-final BN gammas are set to 0.25 (not 0, not 1): every branch stays
observable while the perturbation gain of the random network stays below one.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from tests import util

CONFIGS = {
    'r50v1': dict(resnet_size=50),
    'r50v1-d': dict(resnet_size=50),
    'a-r50': dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                  anti_alias_filter_size=3),
    'a-r50-d': dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                    anti_alias_filter_size=3),
    'a-r152': dict(resnet_size=152, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                   anti_alias_filter_size=3, bl_alpha=1, bl_beta=2),
    # BigLittle with beta = 1: the little branch has as many blocks as the big one (2 / 3 / 5), so both lists of the
    # two-stream backward interleave are long (the published recipes' little branch is one block at depth 50)
    'a-r50-beta1-d': dict(resnet_size=50, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                          anti_alias_filter_size=3, bl_alpha=2, bl_beta=1),
    'se-proj': dict(resnet_size=50, use_se_block=True, anti_alias_type='proj', anti_alias_filter_size=3),
    # the two off-recipe configurations of tests/golden/reference_taps.json: GeM pooling + embedding head on ResNet-101,
    # and no_downsample + flatten pooling + the sigmoid loss's dense-bias initialisation
    'r101v1-gem-emb': dict(resnet_size=101, pool_type='gem', embedding_size=128),
    'r50v1-nodown-flatten-sigmoid': dict(resnet_size=50, no_downsample=True, pool_type='flatten', loss_type='sigmoid'),
}


def uses_d(name):
  return name.endswith('-d')


# ---- committed oracle outputs of the literal-size forward passes ----------------------------------------------------
# The BASELINE configurations at their own sizes (256 / 512 / 128 images at 224 x 224) cost the CPU oracle one to two
# minutes each -- most of the GPU tier's wall time.  Their oracle side depends on nothing but seeds, so it is computed once
# by tests/golden/make_oracle_forward.py (which calls the functions below) and committed as tests/golden/oracle_forward/
# <key>.npz: logits (float32), the scalars, and of every named tap a fixed pseudo-random subset of 65536 elements (the
# rel-L2 over the subset estimates the rel-L2 over the tensor).  ASM_ORACLE_LIVE=1 recomputes instead of loading.
import os as _os

GOLDEN_FWD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'golden', 'oracle_forward')
TAP_SUBSET = 1 << 16


def tap_subset_index(numel, key):
  if numel <= TAP_SUBSET:
    return None
  seed = int.from_bytes(key.encode()[-8:].rjust(8, b'0'), 'little') % (2 ** 31)
  return torch.from_numpy(np.random.default_rng(seed).integers(0, numel, size=TAP_SUBSET))


def tap_subset(t, key):
  idx = tap_subset_index(t.numel(), key)
  flat = t.detach().reshape(-1)
  return flat if idx is None else flat[idx.to(flat.device)]


def oracle_cached(key, compute):
  """-> (dict of numpy arrays, from_cache)"""
  path = _os.path.join(GOLDEN_FWD, key + '.npz') if key is not None else None
  if path is not None and _os.path.exists(path) and _os.environ.get('ASM_ORACLE_LIVE', '0') != '1':
    z = np.load(path)
    return {k: z[k] for k in z.files}, True
  return compute(), False


def make_pair(name, device, batch, size, seed=0, damp=0.25):
  """(oracle model, product model) with identical variables; block-final gammas damped to 0.25 (``damp=None``: left at the
  recipe's literal 0 -- zero_gamma=True, nets/resnet_model.py:84 -- so every residual branch is exactly switched off)."""
  from assembled_cnn_amd.model import Model
  from oracle import assembled_oracle as O
  kw = CONFIGS[name]
  d = uses_d(name)
  om = O.Model(num_classes=1001, emulate_bf16=True, zero_gamma=True, seed=seed, **kw)
  pm = Model(num_classes=1001, device=device, zero_gamma=True, seed=seed, **kw)
  probe = torch.zeros(2, size, size, 3)
  om(probe, True, use_resnet_d=d)
  om.vars.pending_updates = {}
  with torch.no_grad():
    for n, t in om.vars.trainable.items():
      if damp is not None and n.endswith('gamma') and float(t.abs().sum()) == 0:
        t.fill_(damp)
  pm.build((size, size), use_resnet_d=d)
  util.load_oracle_into_product(om, pm)
  return om, pm


def inputs(batch, size, seed=1):
  from oracle import assembled_oracle as O
  img = util.seeded_images(batch, size, size, seed)
  x = img.float() - torch.tensor(O.CHANNEL_MEANS)
  labels = torch.from_numpy(np.random.default_rng(seed + 2).integers(1, 1001, size=batch)).to(torch.int32)
  return img, x, labels


def oracle_forward_record(om, name, batch, size, training):
  """the oracle side of check_forward as a dict of arrays (what tests/golden/make_oracle_forward.py commits)"""
  _, x, _ = inputs(batch, size)
  with torch.no_grad():      # forward only: no autograd graph (batch 256 x 224 x 224 would not fit otherwise)
    lo = om(x, training, use_resnet_d=uses_d(name)).detach()
  rec = {'logits': lo.numpy().astype(np.float32)}
  for k, v in om.taps_nhwc().items():
    rec['tap/' + k] = tap_subset(v.detach().float(), k).numpy().astype(np.float32)
  return rec


def check_forward(name, device, batch, size, training, logits_tol, early_tol=4e-3, stats=None, damp=0.25, golden=None):
  from oracle import assembled_oracle as O
  om, pm = make_pair(name, device, batch, size, damp=damp)
  d = uses_d(name)
  if not training:
    util.perturb_bn_state(om, 7)
    util.load_oracle_into_product(om, pm)
  _, x, labels = inputs(batch, size)
  rec, cached = oracle_cached(golden, lambda: oracle_forward_record(om, name, batch, size, training))
  lo = torch.from_numpy(rec['logits'])
  lp = pm(x.to(device), training, use_resnet_d=d, record_tape=False).float().cpu()
  taps_o = {k[4:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith('tap/')}
  assert 'initial_conv' in pm.taps and 'final_dense' in pm.taps
  e0 = util.rel_l2(tap_subset(pm.taps['initial_conv'].float(), 'initial_conv').cpu(), taps_o['initial_conv'])
  assert e0 <= early_tol, 'initial_conv rel_l2 %.3e' % e0
  for k, v in taps_o.items():
    if k in pm.taps and k != 'final_dense':
      e = util.rel_l2(tap_subset(pm.taps[k].float(), k).cpu(), v)
      assert e <= 2.5 * logits_tol, '%s: tap %s rel_l2 %.3e' % (name, k, e)
  e = util.rel_l2(lp, lo)
  assert e <= logits_tol, '%s logits rel_l2 %.3e > %.1e' % (name, e, logits_tol)
  # top-1 agreement wherever the oracle's top-2 margin is comfortably above the noise
  top2 = lo.topk(2, dim=1).values
  margin = (top2[:, 0] - top2[:, 1]) / lo.std(dim=1)
  sure = margin > 0.3
  assert bool((lp.argmax(1)[sure] == lo.argmax(1)[sure]).all()), 'top-1 mismatch on a large-margin row'
  if stats is not None:
    stats['top1_agree'] = int((lp.argmax(1) == lo.argmax(1)).sum())
    stats['rows'] = int(lp.shape[0])
    oh = F.one_hot(labels.long(), lo.shape[1]).float()
    stats['loss_oracle'] = float(O.softmax_cross_entropy(lo, oh, 0.0))
    stats['loss_product'] = float(O.softmax_cross_entropy(lp, oh, 0.0))
    stats['logits_rel_l2'] = e
    stats['oracle_from_golden'] = cached
  return e


def oracle_noise_floor_record(om, name, batch, size):
  """the oracle side of check_forward_noise_floor: the bf16-emulating forward (logits, tap subsets) and, per tap, how far
  the oracle's own fp32 evaluation is from it (tests/golden/make_oracle_forward.py commits it)"""
  from oracle import assembled_oracle as O
  d = uses_d(name)
  _, x, _ = inputs(batch, size)
  with torch.no_grad():
    lo = om(x, True, use_resnet_d=d).detach()
  taps_o = {k: v.detach().clone() for k, v in om.taps_nhwc().items()}
  of = O.Model(num_classes=1001, emulate_bf16=False, zero_gamma=True, seed=0, **CONFIGS[name])
  of(torch.zeros(2, size, size, 3), True, use_resnet_d=d)
  of.vars.pending_updates = {}
  with torch.no_grad():
    for n, t in om.vars.trainable.items():
      of.vars.trainable[n].copy_(t)
    lf = of(x, True, use_resnet_d=d).detach()
  taps_f = of.taps_nhwc()
  rec = {'logits': lo.numpy().astype(np.float32), 'noise/logits': np.float64(util.rel_l2(lf, lo))}
  for k, v in taps_o.items():
    rec['tap/' + k] = tap_subset(v.float(), k).numpy().astype(np.float32)
    rec['noise/' + k] = np.float64(util.rel_l2(taps_f[k].detach(), v))
  return rec


def check_forward_noise_floor(name, device, batch, size, slack=1.25, floor=4e-3, golden=None):
  """Very deep nets (A-R152: 70 blocks) amplify bf16 rounding at random init beyond any fixed tolerance, so the
  bound is calibrated on the spot: at every named tap the product must be at least as close to the bf16-emulating
  oracle as that oracle is to its own fp32 evaluation (x slack).  A wiring or kernel bug shows up as an error well
  above the rounding noise at the first tap it touches."""
  om, pm = make_pair(name, device, batch, size)
  d = uses_d(name)
  _, x, _ = inputs(batch, size)
  rec, _ = oracle_cached(golden, lambda: oracle_noise_floor_record(om, name, batch, size))
  lo = torch.from_numpy(rec['logits'])
  lp = pm(x.to(device), True, use_resnet_d=d).float().cpu()
  report = {}
  for key in rec:
    if not key.startswith('tap/'):
      continue
    k = key[4:]
    if k not in pm.taps:
      continue
    pv = tap_subset(lp if k == 'final_dense' else pm.taps[k].float(), k).cpu()
    e_p, e_f = util.rel_l2(pv, torch.from_numpy(rec[key])), float(rec['noise/' + k])
    report[k] = (e_p, e_f)
    assert e_p <= max(slack * e_f, floor), '%s: tap %s product-vs-oracle %.3e > %.2f x rounding noise %.3e' % (
        name, k, e_p, slack, e_f)
  assert util.rel_l2(lp, lo) <= max(slack * float(rec['noise/logits']), floor)
  return report


def check_backward(name, device, batch, size, label_smoothing=0.1, min_cos=0.8, min_global_cos=0.9, damp=0.25):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  om, pm = make_pair(name, device, batch, size, damp=damp)
  d = uses_d(name)
  _, x, labels = inputs(batch, size)
  lo = om(x, True, use_resnet_d=d)
  loss = O.softmax_cross_entropy(lo, F.one_hot(labels.long(), 1001).float(), label_smoothing)
  params = list(om.vars.trainable.values())
  og = torch.autograd.grad(loss, params)
  pm(x.to(device), True, use_resnet_d=d)
  oh = ops.onehot(labels.to(device), batch, 1001)
  rows, dz = ops.softmax_ce(pm.logits_padded, pm.ldc, oh, None, batch, 1001, label_smoothing, 0.0, 1.0, pm.ldc)
  pm.backward(dz)
  lp = float(rows.float().mean())
  assert abs(lp - float(loss)) <= 2e-2 * abs(float(loss)), 'loss %.4f vs %.4f' % (lp, float(loss))
  allp, allo = [], []
  for (pname, p), g in zip(om.vars.trainable.items(), og):
    pg = util.product_to_oracle_grad(pname, pm.arena.g(pname).cpu(), p).double().reshape(-1)
    gg = g.double().reshape(-1)
    assert torch.isfinite(pg).all(), pname
    if float(gg.norm()) == 0.0:      # literal zero gamma: nothing flows into a switched-off residual branch
      assert float(pg.norm()) == 0.0, '%s: the oracle gradient is exactly 0, the product gradient has norm %.3e' % (pname, float(pg.norm()))
      continue
    cos = float((pg * gg).sum() / (pg.norm() * gg.norm() + 1e-30))
    ratio = float(pg.norm() / (gg.norm() + 1e-30))
    assert cos >= min_cos, '%s: gradient cosine %.3f (norm ratio %.3f)' % (pname, cos, ratio)
    assert 0.7 <= ratio <= 1.4, '%s: gradient norm ratio %.3f' % (pname, ratio)
    allp.append(pg)
    allo.append(gg)
  allp, allo = torch.cat(allp), torch.cat(allo)
  gcos = float((allp * allo).sum() / (allp.norm() * allo.norm()))
  gratio = float(allp.norm() / allo.norm())
  assert gcos >= min_global_cos, 'global gradient cosine %.3f' % gcos
  assert 0.9 <= gratio <= 1.1, 'global gradient norm ratio %.3f' % gratio
  return gcos


def check_train_steps(name, device, batch, size, steps, hp_kwargs, mixup_type=0, kd_temp=0.0, rel_tol=2e-2, state_tol=2e-2,
                      weight_tol=1e-2, mom_cos=0.85, dec_band=(0.7, 1.3)):
  """A few optimisation steps of the product Trainer vs the oracle's train_step on the same batch:
  loss trajectories must agree and both must decrease."""
  from assembled_cnn_amd.train import HParams, Trainer
  from oracle import assembled_oracle as O
  kw = dict(CONFIGS[name])
  d = uses_d(name)
  hp = HParams(resnet_size=kw.get('resnet_size', 50), resnet_version=kw.get('resnet_version', 1),
               use_sk_block=kw.get('use_sk_block', False), use_se_block=kw.get('use_se_block', False),
               anti_alias_type=kw.get('anti_alias_type', ''), anti_alias_filter_size=kw.get('anti_alias_filter_size', 0),
               bl_alpha=kw.get('bl_alpha', 2), bl_beta=kw.get('bl_beta', 4),
               use_resnet_d=d, zero_gamma=True, mixup_type=mixup_type, kd_temp=kd_temp,
               learning_rate_decay_type='fixed', batch_size=batch, **hp_kwargs)
  tr = Trainer(hp, seed=0, device=device)
  om = O.Model(num_classes=1001, emulate_bf16=True, zero_gamma=True, seed=0, **kw)
  nin = batch * 2 if mixup_type == 1 else batch
  img, _, labels = inputs(nin, size)
  om(torch.zeros(2, size, size, 3), True, use_resnet_d=d)
  om.vars.pending_updates = {}
  with torch.no_grad():
    for n, t in om.vars.trainable.items():
      if n.endswith('gamma') and float(t.abs().sum()) == 0:
        t.fill_(0.25)
  tr.model.build((size, size), use_resnet_d=d)
  util.load_oracle_into_product(om, tr.model)
  rng = np.random.default_rng(4)
  lam1 = torch.from_numpy(rng.beta(0.2, 0.2, size=nin // 2).astype(np.float32)) if mixup_type else None
  lam2 = torch.from_numpy(rng.beta(0.2, 0.2, size=nin // 2).astype(np.float32)) if mixup_type == 2 else None
  if kd_temp > 0:
    teacher_logits = torch.from_numpy(rng.normal(0, 3, size=(nin, 1001)).astype(np.float32))
    lab_o = torch.cat([F.one_hot(labels.long(), 1001).float(), teacher_logits], 1)
    lab_p = lab_o.to(device)
  else:
    lab_o, lab_p = labels, labels.to(device)
  state = O.TrainState(om)
  w_start = {n: t.detach().clone() for n, t in om.vars.trainable.items()}
  x_o = O.mean_image_subtraction(img.float())
  lo_hist, lp_hist = [], []
  for s in range(steps):
    r = O.train_step(state, x_o, lab_o, lr=hp.base_learning_rate, momentum=hp.momentum,
                     weight_decay=hp.weight_decay, label_smoothing=hp.label_smoothing, kd_temp=kd_temp,
                     mixup_type=mixup_type, lam1=lam1, lam2=lam2, use_resnet_d=d)
    lo_hist.append(float(r['parts']['cross_entropy'] + r['parts']['cross_entropy_kd']))
    tr.train_step(img.to(device), lab_p, lam1.to(device) if lam1 is not None else None,
                  lam2.to(device) if lam2 is not None else None)
    lp_hist.append(float(tr.cross_entropy()))
  for a, b in zip(lp_hist, lo_hist):
    assert abs(a - b) <= rel_tol * abs(b), 'loss trajectories diverge: %s vs %s' % (lp_hist, lo_hist)
  assert lp_hist[-1] < lp_hist[0] and lo_hist[-1] < lo_hist[0], 'loss must decrease: %s %s' % (lp_hist, lo_hist)
  dec_p, dec_o = lp_hist[0] - lp_hist[-1], lo_hist[0] - lo_hist[-1]
  assert dec_band[0] <= dec_p / dec_o <= dec_band[1], 'loss decrease %.4f vs oracle %.4f' % (dec_p, dec_o)
  # Post-step STATE (SURVEY section 7 step 1 lists the post-step weights as an oracle output).  Every moving statistic
  # (UPDATE_OPS, nets/run_loop_classification.py:166-178), every fp32 master weight and every momentum slot
  # (nets/optimizer_setting.py:29-37) against the oracle's after the same steps.  A wrong BN momentum on one path (the SK
  # unit's small-batch BN, the dual shortcut BN), a variable missing from the weight-decay set or from the update would show
  # here and nowhere in the loss trajectory.  Moving statistics are averages of batch statistics (rounding noise does not
  # accumulate: rel-L2 <= 2e-2 each); the UPDATE of a weight is lr * (a few noisy gradients), so weights are compared as
  # weights (rel-L2 <= 1e-2: the update itself is ~1e-3 of the weight) and momentum slots by cosine / norm like gradients.
  a = tr.model.arena
  worst_state = ('', 0.0)
  grow = 1.0 - float(hp.bn_momentum) ** steps       # how far a moving statistic has moved from its initial value
  for n, t in om.vars.state.items():
    e = util.rel_l2(a.st(n).cpu(), t.detach())
    if n.endswith('moving_mean'):
      # a moving mean is (1 - momentum^steps) x an average of batch means, which may sit near 0 while its rounding noise
      # scales with the channel's standard deviation: measure the difference against that scale when it is the larger one
      sd = om.vars.state[n[:-len('moving_mean')] + 'moving_variance'].detach().double().clamp_min(0).sqrt()
      den = max(float(t.detach().double().norm()), grow * float(sd.norm()))
      e = float((a.st(n).cpu().double() - t.detach().double()).norm()) / (den if den > 0 else 1.0)
    worst_state = max(worst_state, (n, e), key=lambda v: v[1])
    assert e <= state_tol, 'moving statistic %s rel_l2 %.3e after %d steps' % (n, e, steps)
  assert state.accums is not None and len(state.accums) == len(om.vars.trainable)
  worst_w = ('', 0.0)
  all_mp, all_mo = [], []
  for (n, t), acc in zip(om.vars.trainable.items(), state.accums):
    wp = util.product_to_oracle_grad(n, a.w(n).float().cpu(), t)
    e = util.rel_l2(wp, t.detach())
    worst_w = max(worst_w, (n, e), key=lambda v: v[1])
    # ... or, where the update itself is large against the weight (the stem at a tiny batch: the gradient at the end of
    # the backward chain is the noisiest), as an UPDATE: direction and size of (w - w_start) like a gradient in check_backward
    if e > weight_tol:
      up, uo = (wp.double() - w_start[n].double()).reshape(-1), (t.detach().double() - w_start[n].double()).reshape(-1)
      cos = float((up * uo).sum() / (up.norm() * uo.norm() + 1e-30))
      ratio = float(up.norm() / (uo.norm() + 1e-30))
      if up.numel() >= 4096:
        assert cos >= 0.5 and 0.5 <= ratio <= 2.0, 'master weight %s rel_l2 %.3e; its update: cosine %.3f, norm ratio %.3f' % (
            n, e, cos, ratio)
      else:     # a 64 .. 2048-element gamma / beta vector at batch 4 - 8: the update's direction is noise, its size is not
        assert 0.4 <= ratio <= 2.5, 'master weight %s rel_l2 %.3e; its update: cosine %.3f, norm ratio %.3f' % (n, e, cos, ratio)
    mp_ = util.product_to_oracle_grad(n, a.m(n).float().cpu(), t).double().reshape(-1)
    mo = acc.detach().double().reshape(-1)
    all_mp.append(mp_)
    all_mo.append(mo)
    if float(mo.norm()) > 0:
      cos = float((mp_ * mo).sum() / (mp_.norm() * mo.norm() + 1e-30))
      ratio = float(mp_.norm() / mo.norm())
      # (per variable only the size, and only for tensors large enough for a norm to mean something: at batch 4 - 8 the squeeze
      # layers' batch norm over that many rows and the 64-element gamma / beta vectors are noise as directions; the direction is
      # checked over all slots together below)
      if mp_.numel() >= 4096:
        assert 0.5 <= ratio <= 2.0, 'momentum slot %s: cosine %.3f norm ratio %.3f' % (n, cos, ratio)
    else:
      assert float(mp_.norm()) == 0.0, 'momentum slot %s should be zero' % n
  all_mp, all_mo = torch.cat(all_mp), torch.cat(all_mo)
  gcos = float((all_mp * all_mo).sum() / (all_mp.norm() * all_mo.norm() + 1e-30))
  gratio = float(all_mp.norm() / (all_mo.norm() + 1e-30))
  assert gcos >= mom_cos and 0.85 <= gratio <= 1.15, 'all momentum slots: cosine %.3f norm ratio %.3f' % (gcos, gratio)
  print('post-step state: worst moving statistic %s %.3e, worst master weight %s %.3e, momentum cosine %.3f ratio %.3f' % (
      worst_state + worst_w + (gcos, gratio)))
  return lp_hist, lo_hist


def check_teacher_forced(name, device, batch, size, training=True, out_tol=4e-3, in_tol=1e-2, squeeze_tol=2e-2):
  """Per-layer parity over the WHOLE network without depth amplification.

  The oracle (bf16-emulating) records the input of every convolution and the output (and residual operand) of every
  fused conv -> BN [-> + residual] [-> ReLU] group.  The product then runs the same forward pass, but every such
  group is FED THE ORACLE'S INPUT (teacher forcing), so each comparison sees one product layer on identical bf16
  inputs -- the per-kernel tolerance (rel-L2 <= 4e-3) applies at layer 150 just as at layer 1:
    * group output vs the oracle's                                      -> out_tol
      (squeeze layers -- SK / SE fc on [N,1,1,d], batch statistics over N values only -- squeeze_tol);
    * the tensor the product itself computed for that group's input (from the previous forced group through its own
      pooling / blur / SK gap + select / SE / upsample-add kernels) vs the oracle's input, BEFORE it is replaced, and
      the same for the residual operand                                 -> in_tol
  Returns the list of (layer, kind, error) sorted by error."""
  from assembled_cnn_amd import model as pmodel, nn as pnn, ops
  om, pm = make_pair(name, device, batch, size)
  d = uses_d(name)
  if not training:
    util.perturb_bn_state(om, 7)
    util.load_oracle_into_product(om, pm)
  _, x, _ = inputs(batch, size)
  with torch.no_grad():
    lo = om(x, training, use_resnet_d=d, record_layers=True).detach()
  rec_in, rec_bn = om.layer_record
  errs = []

  def to_dev(t_nchw):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device)

  def cmp(got_nhwc, ref_nchw, layer, kind):
    ref = ref_nchw.permute(0, 2, 3, 1)
    e = util.rel_l2(got_nhwc.float().cpu().reshape(ref.shape), ref)
    errs.append((layer, kind, e, tuple(ref.shape)))

  orig = pnn.conv_bn

  def forced(ctx, xv, conv, bn, stride, relu, residual=None, res_mode=0, tap_pre=None):
    if ctx.dry:
      return orig(ctx, xv, conv, bn, stride, relu, residual, res_mode, tap_pre)
    ref_in = rec_in[conv.name]
    if conv.stem:
      cmp(xv.data[:, 3:-3, 3:-3, :3], ref_in, conv.name, 'input')
    else:
      cmp(xv.data, ref_in, conv.name, 'input')
      xv = pnn.Var(to_dev(ref_in), needs_grad=False)
    ref_out, ref_res = rec_bn[bn.gamma]
    if residual is not None:
      assert ref_res is not None, bn.gamma
      cmp(residual.data, ref_res, bn.gamma, 'residual')
      residual = pnn.Var(to_dev(ref_res), needs_grad=False)
    out = orig(ctx, xv, conv, bn, stride, relu, residual, res_mode, tap_pre)
    cmp(out.data, ref_out, bn.gamma, 'squeeze-output' if (ref_out.shape[2] * ref_out.shape[3] == 1) else 'output')
    return out

  # the fused SK unit never materialises its normalised 3x3 output: force its input; its pooled vector is checked as
  # the forced input of sk_fc_1 and its selected output V at the input of the block's last 1x1 convolution
  fused_groups = []
  orig_sk = pnn.SKUnit._call_fused

  def forced_sk(self, ctx, xv, stride):
    ref_in = rec_in[self.conv.name]
    cmp(xv.data, ref_in, self.conv.name, 'input')
    fused_groups.append(self.bn.gamma)
    return orig_sk(self, ctx, pnn.Var(to_dev(ref_in), needs_grad=False), stride)

  pnn.conv_bn = forced
  pmodel.conv_bn = forced
  pnn.SKUnit._call_fused = forced_sk
  try:
    lp = pm(x.to(device), training, use_resnet_d=d, record_tape=False).float().cpu()
  finally:
    pnn.conv_bn = orig
    pmodel.conv_bn = orig
    pnn.SKUnit._call_fused = orig_sk
  # the head: GAP + dense on the (product-computed) output of the last forced group
  e_logits = util.rel_l2(lp, lo)
  errs.append(('final_dense', 'logits', e_logits, tuple(lo.shape)))
  n_groups = sum(1 for e in errs if e[1] in ('output', 'squeeze-output'))
  assert n_groups + len(fused_groups) == len(rec_bn), 'forced %d + %d groups, the oracle recorded %d' % (
      n_groups, len(fused_groups), len(rec_bn))
  lim = {'output': out_tol, 'squeeze-output': squeeze_tol, 'input': in_tol, 'residual': in_tol, 'logits': in_tol}
  errs.sort(key=lambda t: -t[2] / lim[t[1]])
  bad = [t for t in errs if not t[2] <= lim[t[1]]]
  assert not bad, '%s: %d of %d teacher-forced comparisons out of tolerance; worst: %s' % (
      name, len(bad), len(errs), ['%s %s %.3e %s' % t for t in bad[:8]])
  return errs


def check_teacher_forced_backward(name, device, batch, size, label_smoothing=0.1, keep_prob=1.0, dx_tol=6e-3,
                                  lazy_tol=8e-3, dparam_tol=6e-3, dw_tol=6e-3, squeeze_tol=2e-2, sk_tol=3e-2, env=None,
                                  capture=None):
  """Per-layer parity of the hand-written BACKWARD tape over the whole network, without depth amplification.

  The bf16-emulating oracle runs forward + autograd backward once and keeps, for every conv -> BN [-> + residual]
  [-> ReLU] group, its input, its output and d loss / d (both) (``record_live``).  The product then runs
    * its forward with every group fed the ORACLE'S input (as check_teacher_forced does -- but by overriding the
      tensor of the SAME activation object, so the tape stays connected and every fused / lazy path of nn.py runs
      exactly as in training: deferred + dual batch norm of a projection shortcut, lazily masked shortcut / merge
      gradients, pooled gradient gathered by conv1's input gradient, the fused SK unit, the reordered projection tape);
    * its backward tape with every group's closure wrapped: when the closure is about to run, the gradient the product
      has ACCUMULATED for the group's output (all fan-in terms the tape produced) is compared with the oracle's
      d loss / d output, and then REPLACED by it (teacher forcing), so that the next comparison again sees one group's
      worth of product kernels on the oracle's operands -- at layer 150 as at layer 1.
  A gradient still held in lazy form (dy + ReLU mask, + pooled contribution) is compared in materialised form but left
  as it is (its operands were forced one group upstream); a pre-computed batch-norm backward (dual path) is checked
  through its effects (dx of the shortcut convolution at the next forced point, dW / dgamma / dbeta).
  Tolerance classes (measured on MI355X, batch 16): bf16 gradient storage alone costs 2-3e-3 per comparison (the forced
  gradient, dy and dx are each rounded to 8 bits), so dout / dx / dW / dgamma / dbeta sit at 2-4e-3 against the 6e-3 bound.
  Three documented noise classes are wider: (1) '-squeeze' -- tensors of the [N,1,1,d] layers, whose batch norm runs over
  the N pooled vectors only and whose backward is a difference of nearly equal terms: 2e-2, and 1.5e-1 for the 32..256-row
  kernel gradients of sk_fc_1 / sk_fc_2 / the SE pair (measured up to 1.05e-1 at 224 x 224, 2.9e-2 at 64 x 64; a dropped
  term or a wrong sign is O(1)); (2) '-sk' -- gamma / beta of the SK unit's 3x3 batch norm, which inherit that path through
  dU = ds / HW summed over H*W pixels: 3e-2; (3) 'dbeta-maxpool' -- see the comment at its definition.
  Comparisons: 'dout' accumulated output gradient of a group at a forced point, 'dout-lazy' the same in lazy form,
  'dx' the input gradient of a convolution whose input is not itself a group output (pooled / blurred / SK tensors),
  'dW', 'dgamma', 'dbeta', 'dbias' every trainable variable's gradient.  Returns the list (layer, kind, error, shape)."""
  import os
  from assembled_cnn_amd import model as pmodel, nn as pnn, ops
  from oracle import assembled_oracle as O
  old_env = {}
  for k, v in (env or {}).items():
    old_env[k] = os.environ.get(k)
    os.environ[k] = v
  ops.refresh_tuning()
  try:
    om, pm = make_pair(name, device, batch, size)
    d = uses_d(name)
    _, x, labels = inputs(batch, size)
    training_kp = float(keep_prob)
    uniforms_o = uniforms_p = None
    if training_kp < 1.0:
      gen = torch.Generator().manual_seed(11)
      drawn = []

      def draw(shape):      # oracle asks [1, C, H-6, W-6]; the product wants [H-6, W-6, C]
        u = torch.rand(shape, generator=gen)
        drawn.append(u)
        return u
      uniforms_o = draw
    lo = om(x, True, use_resnet_d=d, keep_prob=training_kp, dropblock_uniforms=uniforms_o, record_layers=True,
            record_live=True)
    if training_kp < 1.0:
      uniforms_p = [u[0].permute(1, 2, 0).contiguous().to(device) for u in drawn]
    rec_in, rec_bn = om.layer_record
    rec_extra = om.extra_record
    if capture is not None:      # debugging aid (tools/debug): the oracle's records
      capture.update(om=om, pm=pm, rec_in=rec_in, rec_bn=rec_bn, rec_extra=rec_extra)
    loss = O.softmax_cross_entropy(lo, F.one_hot(labels.long(), 1001).float(), label_smoothing)
    loss.backward()
    errs = []
    n_forced = [0]

    def to_dev(t_nchw):
      return t_nchw.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device)

    def cmp(got_nhwc, ref_nchw, layer, kind):
      ref = ref_nchw.detach().permute(0, 2, 3, 1)
      e = util.rel_l2(got_nhwc.float().cpu().reshape(ref.shape), ref)
      errs.append((layer, kind, e, tuple(ref.shape)))

    def peek_grad(v):
      """the gradient a Var holds, materialised WITHOUT touching its lazy state; (tensor or None, was_lazy)"""
      g, lazy = v._grad, False
      if g is not None and v.grad_mask is not None:
        g, lazy = ops.mask_apply(g, v.grad_mask), True
      if v.pool_grad is not None:
        dp, k, stride, pad, cv = v.pool_grad
        s = ops.avgpool_bwd(dp, v.shape, k, stride, pad, cv)
        g, lazy = (s if g is None else ops.add_bf16(g, s)), True
      return g, lazy

    group_outputs = {}

    maxpool_fed = [None]
    pending_dx = {}     # id(conv input that is not a forced point) -> [var, oracle tensor, label, consumers still to run]
    last_gamma = [None]

    def wrap_last_closure(ctx, out, ref_out, label, squeeze, x_var=None, ref_x=None, x_label=None):
      if ctx.tape is None or not ctx.tape:
        return
      inner = ctx.tape[-1]
      group_outputs[id(out)] = True
      if x_var is not None and id(x_var) not in group_outputs and x_var.needs_grad:
        ent = pending_dx.setdefault(id(x_var), [x_var, ref_x, x_label, 0])
        ent[3] += 1

      def bwd():
        if out.pre_dy is None and ref_out.grad is not None:
          g, lazy = peek_grad(out)
          assert g is not None, '%s: no gradient reached this group' % label
          cmp(g, ref_out.grad, label, 'dout-squeeze' if squeeze else ('dout-lazy' if lazy else 'dout'))
          if not lazy:
            out._grad, out.grad_mask, out.grad_owned = to_dev(ref_out.grad), None, True
            n_forced[0] += 1
        inner()
        ent = pending_dx.get(id(x_var)) if x_var is not None else None
        if ent is not None:
          ent[3] -= 1
          if ent[3] == 0 and ent[1].grad is not None:     # every convolution reading this tensor has run its backward
            gx, _ = peek_grad(x_var)
            if gx is not None:
              cmp(gx, ent[1].grad, ent[2], 'dx-squeeze' if ent[1].shape[2] * ent[1].shape[3] == 1 else 'dx')
      ctx.tape[-1] = bwd

    # The saved batch-norm coefficients are teacher-forced too.  The product derives them from the bf16 conv output's
    # fp32 partial sums, the oracle from a two-pass fp32 mean / variance: they agree to ~5e-5 of a standard deviation, but
    # that difference is SYSTEMATIC per channel, so every element of the channel within 5e-5 sigma of the ReLU threshold
    # flips its mask -- and a fraction p of flipped masks moves a masked gradient by sqrt(p) (4e-5 -> 6e-3, as much as the
    # whole tolerance).  With the oracle's coefficients the masks agree except at exact float32 ties.
    cur = {}
    orig_fin = ops.bn_finalize

    def forced_finalize(part, M, Cn, gamma, beta, eps, momentum, mm, mv):
      mean, invstd, scale, shift = orig_fin(part, M, Cn, gamma, beta, eps, momentum, mm, mv)
      y = cur.pop('y', None)
      if y is not None:
        yd = y.detach().double()
        red = [i for i in range(yd.dim()) if i != 1]
        mu, var = yd.mean(red), yd.var(red, unbiased=False)
        isd = 1.0 / torch.sqrt(var + eps)
        sc = gamma.detach().double().cpu() * isd
        sh = beta.detach().double().cpu() - mu * sc
        for dst, src in ((mean, mu), (invstd, isd), (scale, sc), (shift, sh)):
          dst.copy_(src.float().to(dst.device))
      return mean, invstd, scale, shift

    orig = pnn.conv_bn

    def forced(ctx, xv, conv, bn, stride, relu, residual=None, res_mode=0, tap_pre=None):
      if ctx.dry:
        return orig(ctx, xv, conv, bn, stride, relu, residual, res_mode, tap_pre)
      ref_in = rec_in[conv.name]
      cur['y'] = rec_extra.get('conv_out:' + conv.name)
      if not conv.stem:
        xv._data = to_dev(ref_in)
      ref_out, ref_res = rec_bn[bn.gamma]
      if residual is not None and not (residual._data is None and residual.deferred is not None):
        residual._data = to_dev(ref_res)
      ref_y = rec_extra.get('conv_out:' + conv.name) if ref_out.shape[2] * ref_out.shape[3] == 1 else None
      if ref_y is not None:
        # sk_fc_1 (+ batch norm over the N pooled vectors only): at random init the pooled features barely vary across
        # images, |mean| / std of the pre-activation is 50-100, so ONE bf16 rounding flip of one pre-activation (2^-8 of
        # the mean; a 1e-6 summation-order difference flips ~3e-4 of them) moves that channel's normalised values -- and
        # its 16 ReLU decisions -- by tenths of a standard deviation.  The saved pre-activation is therefore forced too
        # (the forward teacher-forced check covers the product's own value of it).
        y_forced = to_dev(ref_y)
        conv.fprop = lambda d_, x_, ws_: (y_forced, None)
      try:
        out = orig(ctx, xv, conv, bn, stride, relu, residual, res_mode, tap_pre)
      finally:
        cur.pop('y', None)
        if ref_y is not None:
          del conv.fprop
      if ref_y is not None and relu and ctx.tape:
        # ... and so is the saved ReLU mask.  bf16 pre-activations sit on a coarse grid, so over a batch of 8-16 one of
        # them regularly EQUALS the channel mean: its normalised value is 0 up to float32 rounding (|out| <= 1e-7 on both
        # sides, measured), and whether ReLU'(0) counts as 0 or 1 is decided by the last bit.  One such tie is 3-7 % of a
        # squeeze layer's dbeta (16 terms per channel).  The oracle's decisions are written into the product's packed mask.
        fn = ctx.tape[-1]
        cell = fn.__closure__[fn.__code__.co_freevars.index('mask_t')].cell_contents
        if cell is not None:
          bits = (ref_out.detach().reshape(ref_out.shape[0], -1) > 0).to(torch.uint8).view(ref_out.shape[0], -1, 8)
          packed = (bits * (2 ** torch.arange(8, dtype=torch.uint8))).sum(-1).to(torch.uint8)
          cell.copy_(packed.to(cell.device).view(cell.shape))
      last_gamma[0] = bn.gamma
      if tap_pre == 'initial_conv' and pm.resnet_version == 1:
        maxpool_fed[0] = bn.gamma
      squeeze = ref_out.shape[2] * ref_out.shape[3] == 1
      wrap_last_closure(ctx, out, ref_out, bn.gamma, squeeze, None if conv.stem else xv, ref_in, conv.name)
      return out

    # the SE / DropBlock form of a block ends in a separate add (+ ReLU): its output is a forced point as well
    orig_add = pmodel.Model._add_relu

    def forced_add(ctx, a_, b_, relu):
      out = orig_add(ctx, a_, b_, relu)
      if not ctx.dry:
        key = 'block_out:' + last_gamma[0]
        # its ReLU mask is read from the output at backward time: use the oracle's tensor, as every block whose output
        # feeds a convolution does anyway (the separate bf16 add rounds once more than the oracle, and a sign flip of an
        # element at ~0 moves the masked gradient by sqrt(fraction flipped))
        out._data = to_dev(rec_extra[key])
        wrap_last_closure(ctx, out, rec_extra[key], key, False)
      return out

    orig_sk = pnn.SKUnit._call_fused

    def forced_sk(self, ctx, xv, stride):
      xv._data = to_dev(rec_in[self.conv.name])
      cur['y'] = rec_extra.get('conv_out:' + self.conv.name)
      v = orig_sk(self, ctx, xv, stride)
      wrap_last_closure(ctx, v, rec_extra['sk_out:' + self.bn.gamma], 'sk_out:' + self.bn.gamma, False)
      return v

    pnn.conv_bn = forced
    pmodel.conv_bn = forced
    pnn.SKUnit._call_fused = forced_sk
    pmodel.Model._add_relu = staticmethod(forced_add)
    ops.bn_finalize = forced_finalize
    try:
      pm(x.to(device), True, use_resnet_d=d, keep_prob=training_kp, dropblock_uniforms=uniforms_p)
      # the loss layer is teacher-forced too: d loss / d logits from the ORACLE's logits through the product's kernel
      lpad = torch.zeros((batch, pm.ldc), dtype=torch.float32)
      lpad[:, :1001] = lo.detach()
      oh = ops.onehot(labels.to(device), batch, 1001)
      rows, dz = ops.softmax_ce(lpad.to(device).view(batch, 1, 1, pm.ldc), pm.ldc, oh, None, batch, 1001, label_smoothing,
                                0.0, 1.0, pm.ldc)
      assert abs(float(rows.float().mean()) - float(loss.detach())) <= 1e-4 * abs(float(loss.detach()))
      pm.backward(dz)
    finally:
      pnn.conv_bn = orig
      pmodel.conv_bn = orig
      pnn.SKUnit._call_fused = orig_sk
      pmodel.Model._add_relu = staticmethod(orig_add)
      ops.bn_finalize = orig_fin
    if device != 'cpu':
      torch.cuda.synchronize()
    # every trainable variable's gradient (each is written by exactly one closure, on operands forced as above)
    sk_bn = set(k[len('sk_out:'):-len('gamma')] for k in rec_extra if k.startswith('sk_out:'))
    for pname, p in om.vars.trainable.items():
      pg = util.product_to_oracle_grad(pname, pm.arena.g(pname).float().cpu(), p).reshape(p.shape)
      leaf = pname.rsplit('/', 1)[1]
      kind = 'dW' if leaf == 'kernel' else 'd' + leaf
      if kind == 'dW' and ('sk_fc' in pname or 'seblock' in pname):
        kind = 'dW-squeeze'
      elif leaf in ('gamma', 'beta') and pname[:-len(leaf)] in sk_bn:
        kind += '-sk'       # the SK unit's 3x3 batch norm: its gradient carries the attention (squeeze) path's noise
      elif leaf in ('gamma', 'beta') and pname[:-len(leaf)] + 'gamma' in rec_bn and \
          rec_bn[pname[:-len(leaf)] + 'gamma'][0].shape[2] * rec_bn[pname[:-len(leaf)] + 'gamma'][0].shape[3] == 1:
        kind += '-squeeze'
      assert torch.isfinite(pg).all(), pname
      if leaf == 'beta' and pname[:-len(leaf)] + 'gamma' == maxpool_fed[0]:
        # The batch norm in front of the max pool (resnet_version 1).  Every consumer of the pooled tensor is a 1x1
        # convolution followed by a batch norm, whose backward makes sum_pixels(dy) = 0 and hence sum_pixels(W^T dy) = 0:
        # the pooled gradient sums to zero per channel, and so does the gradient the max pool routes to its arg-maxima --
        # EXCEPT the share routed to windows whose maximum is exactly 0 (all-negative windows after the ReLU, masked out).
        # dbeta is minus that small subset sum; one window whose maximum is +tiny on one side and 0 on the other (a bf16
        # rounding at the ReLU threshold) moves it by a whole term.  Measured 2-5e-2; a dropped fan-in term would be O(1).
        kind = 'dbeta-maxpool'
      e = util.rel_l2(pg, p.grad)
      errs.append((pname, kind, e, tuple(p.shape)))
    lim = {'dout': dx_tol, 'dx': dx_tol, 'dout-lazy': lazy_tol, 'dout-squeeze': squeeze_tol, 'dx-squeeze': squeeze_tol,
           'dW': dw_tol, 'dW-squeeze': 7.5 * squeeze_tol, 'dgamma': dparam_tol, 'dbeta': dparam_tol, 'dbias': dparam_tol,
           'dgamma-sk': sk_tol, 'dbeta-sk': sk_tol, 'dgamma-squeeze': squeeze_tol, 'dbeta-squeeze': squeeze_tol,
           'dbeta-maxpool': 8e-2}
    errs.sort(key=lambda t: -t[2] / lim[t[1]])
    bad = [t for t in errs if not t[2] <= lim[t[1]]]
    assert not bad, '%s: %d of %d teacher-forced backward comparisons out of tolerance; worst: %s' % (
        name, len(bad), len(errs), ['%s %s %.3e %s' % t for t in bad[:10]])
    stats = {'forced': n_forced[0], 'kinds': {k: sum(1 for e in errs if e[1] == k) for k in lim}}
    return errs, stats
  finally:
    for k, v in old_env.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v
    ops.refresh_tuning()


def _train_forward_oracle_inputs(name, n_in, size, mixup_type, kd_temp):
  """oracle model (variables as every whole-model test sets them) and the seeded inputs of check_train_forward_at_size"""
  from oracle import assembled_oracle as O
  kw = dict(CONFIGS[name])
  d = uses_d(name)
  om = O.Model(num_classes=1001, emulate_bf16=True, zero_gamma=True, seed=0, **kw)
  om(torch.zeros(2, size, size, 3), True, use_resnet_d=d)
  om.vars.pending_updates = {}
  with torch.no_grad():
    for n, t in om.vars.trainable.items():
      if n.endswith('gamma') and float(t.abs().sum()) == 0:
        t.fill_(0.25)
  img, _, labels = inputs(n_in, size)
  rng = np.random.default_rng(4)
  lam1 = torch.from_numpy(rng.beta(0.2, 0.2, size=n_in // 2).astype(np.float32)) if mixup_type else None
  lam2 = torch.from_numpy(rng.beta(0.2, 0.2, size=n_in // 2).astype(np.float32)) if mixup_type == 2 else None
  onehot = F.one_hot(labels.long(), 1001).float()
  tl = torch.from_numpy(rng.normal(0, 3, size=(n_in, 1001)).astype(np.float32)) if kd_temp > 0 else None
  onehot_o, teacher_o = (O.split_kd_labels(torch.cat([onehot, tl], 1), kd_temp) if kd_temp > 0 else (onehot, None))
  x_o = O.mean_image_subtraction(img.float())
  if mixup_type == 1:
    x_o, onehot_o, teacher_o = O.mixup(x_o, onehot_o, lam1, keep_batch_size=False, y_t=teacher_o)
  elif mixup_type == 2:
    x_o, onehot_o, teacher_o = O.mixup(x_o, onehot_o, lam1, keep_batch_size=True, y_t=teacher_o, lam2=lam2)
  return dict(kw=kw, d=d, om=om, img=img, labels=labels, lam1=lam1, lam2=lam2, onehot=onehot, tl=tl, x_o=x_o,
              onehot_o=onehot_o, teacher_o=teacher_o)


def oracle_train_forward_record(name, n_in, size, mixup_type=0, label_smoothing=0.0, kd_temp=0.0, noise_floor=False, setup=None):
  """the oracle side of check_train_forward_at_size as a dict of arrays (tests/golden/make_oracle_forward.py commits it)"""
  from oracle import assembled_oracle as O
  su = setup if setup is not None else _train_forward_oracle_inputs(name, n_in, size, mixup_type, kd_temp)
  om, d, x_o = su['om'], su['d'], su['x_o']
  with torch.no_grad():
    lo = om(x_o, True, use_resnet_d=d).detach()
    loss = float(O.softmax_cross_entropy(lo, su['onehot_o'], label_smoothing) +
                 (O.kd_loss(lo, su['teacher_o'], kd_temp) if kd_temp > 0 else 0.0))
  rec = {'logits': lo.numpy().astype(np.float32), 'loss': np.float64(loss)}
  if noise_floor:
    of = O.Model(num_classes=1001, emulate_bf16=False, zero_gamma=True, seed=0, **su['kw'])
    of(torch.zeros(2, size, size, 3), True, use_resnet_d=d)
    of.vars.pending_updates = {}
    with torch.no_grad():
      for n, t in om.vars.trainable.items():
        of.vars.trainable[n].copy_(t)
      lf = of(x_o, True, use_resnet_d=d).detach()
    rec['noise'] = np.float64(util.rel_l2(lf, lo))
  return rec


def check_train_forward_at_size(name, device, n_in, size, mixup_type=0, label_smoothing=0.0, kd_temp=0.0, logits_tol=6e-2,
                                loss_tol=2e-2, noise_floor=False, slack=1.25, golden=None):
  """The FORWARD half of one training step of a BASELINE configuration at its own per-GPU shard size, product vs oracle:
  raw uint8 images -> [mixup] + mean subtraction (fused kernel) -> network in training mode (batch statistics over the
  whole shard) -> softmax cross entropy [+ label smoothing] [+ KD].  Forward only on the host (the autograd graph of a
  batch-256 step does not fit a host budget); the backward tape is covered per layer by check_teacher_forced_backward.
  Compared: the mixed network input, the mixed targets, logits (rel-L2) and the loss.  ``noise_floor``: for the 70-block
  A-R152 the logits bound is calibrated on the spot (the bf16 oracle vs its own fp32 evaluation, x slack).  ``golden``:
  the oracle's logits / loss / noise figure from tests/golden/oracle_forward/<golden>.npz when that file exists (the mixed
  input and targets are always recomputed here: they are cheap)."""
  from assembled_cnn_amd import ops
  from assembled_cnn_amd.train import HParams, Trainer
  su = _train_forward_oracle_inputs(name, n_in, size, mixup_type, kd_temp)
  kw, d, om, img, labels, lam1, lam2 = su['kw'], su['d'], su['om'], su['img'], su['labels'], su['lam1'], su['lam2']
  x_o, onehot_o, teacher_o = su['x_o'], su['onehot_o'], su['teacher_o']
  batch = n_in // 2 if mixup_type == 1 else n_in
  hp = HParams(resnet_size=kw.get('resnet_size', 50), resnet_version=kw.get('resnet_version', 1),
               use_sk_block=kw.get('use_sk_block', False), use_se_block=kw.get('use_se_block', False),
               anti_alias_type=kw.get('anti_alias_type', ''), anti_alias_filter_size=kw.get('anti_alias_filter_size', 0),
               bl_alpha=kw.get('bl_alpha', 2), bl_beta=kw.get('bl_beta', 4), use_resnet_d=d, zero_gamma=True,
               mixup_type=mixup_type, kd_temp=kd_temp, label_smoothing=label_smoothing, batch_size=batch)
  tr = Trainer(hp, seed=0, device=device)
  tr.model.build((size, size), use_resnet_d=d)
  util.load_oracle_into_product(om, tr.model)
  lab_p = torch.cat([su['onehot'], su['tl']], 1).to(device) if kd_temp > 0 else labels.to(device)
  # ---- product ----
  x_p, oh_p, t_p = tr.prepare_inputs(img.to(device), lab_p, lam1.to(device) if lam1 is not None else None,
                                     lam2.to(device) if lam2 is not None else None)
  lp = tr.model(x_p, True, use_resnet_d=d, prepadded=True, record_tape=False).float().cpu()
  m = tr.model
  rows, _ = ops.softmax_ce(m.logits_padded, m.ldc, oh_p, t_p, batch, 1001, label_smoothing, kd_temp, 1.0, m.ldc, want_grad=False)
  loss_p = float(rows.float().mean())
  # ---- oracle ----
  rec, cached = oracle_cached(golden, lambda: oracle_train_forward_record(name, n_in, size, mixup_type, label_smoothing, kd_temp,
                                                                         noise_floor, setup=su))
  lo, loss_o = torch.from_numpy(rec['logits']), float(rec['loss'])
  report = {'batch': batch, 'oracle_from_golden': cached}
  report['input'] = util.rel_l2(x_p[:, 3:-3, 3:-3, :3].float().cpu(), x_o)
  assert report['input'] <= 4e-3, 'mixed / mean-subtracted network input rel_l2 %.3e' % report['input']
  assert float(x_p[:, :3].float().abs().max()) == 0.0 and float(x_p[..., 3].float().abs().max()) == 0.0, 'halo must be zero'
  report['targets'] = util.max_abs(oh_p.cpu().view(batch, -1)[:, :1001], onehot_o)
  assert report['targets'] <= 1e-6
  if t_p is not None:
    assert util.max_abs(t_p.cpu().view(batch, -1)[:, :1001], teacher_o) <= 1e-5
  e = util.rel_l2(lp, lo)
  report['logits'] = e
  if noise_floor:
    report['noise'] = float(rec['noise'])
    assert e <= max(slack * report['noise'], 4e-3), 'logits: product-vs-oracle %.3e > %.2f x rounding noise %.3e' % (
        e, slack, report['noise'])
  else:
    assert e <= logits_tol, '%s logits rel_l2 %.3e > %.1e' % (name, e, logits_tol)
  report['loss'] = (loss_p, loss_o)
  assert abs(loss_p - loss_o) <= loss_tol * abs(loss_o), 'loss %.5f vs oracle %.5f' % (loss_p, loss_o)
  return report
