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
    'se-proj': dict(resnet_size=50, use_se_block=True, anti_alias_type='proj', anti_alias_filter_size=3),
}


def uses_d(name):
  return name.endswith('-d')


def make_pair(name, device, batch, size, seed=0):
  """(oracle model, product model) with identical variables; block-final gammas damped to 0.25."""
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
      if n.endswith('gamma') and float(t.abs().sum()) == 0:
        t.fill_(0.25)
  pm.build((size, size), use_resnet_d=d)
  util.load_oracle_into_product(om, pm)
  return om, pm


def inputs(batch, size, seed=1):
  from oracle import assembled_oracle as O
  img = util.seeded_images(batch, size, size, seed)
  x = img.float() - torch.tensor(O.CHANNEL_MEANS)
  labels = torch.from_numpy(np.random.default_rng(seed + 2).integers(1, 1001, size=batch)).to(torch.int32)
  return img, x, labels


def check_forward(name, device, batch, size, training, logits_tol, early_tol=4e-3, stats=None):
  from oracle import assembled_oracle as O
  om, pm = make_pair(name, device, batch, size)
  d = uses_d(name)
  if not training:
    util.perturb_bn_state(om, 7)
    util.load_oracle_into_product(om, pm)
  _, x, labels = inputs(batch, size)
  with torch.no_grad():      # forward only: no autograd graph (batch 256 x 224 x 224 would not fit otherwise)
    lo = om(x, training, use_resnet_d=d).detach()
  lp = pm(x.to(device), training, use_resnet_d=d, record_tape=False).float().cpu()
  taps_o = om.taps_nhwc()
  assert 'initial_conv' in pm.taps and 'final_dense' in pm.taps
  e0 = util.rel_l2(pm.taps['initial_conv'].float().cpu(), taps_o['initial_conv'].detach())
  assert e0 <= early_tol, 'initial_conv rel_l2 %.3e' % e0
  for k, v in taps_o.items():
    if k in pm.taps and k != 'final_dense':
      pv = pm.taps[k].float().cpu().reshape(v.shape)
      e = util.rel_l2(pv, v.detach())
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
  return e


def check_forward_noise_floor(name, device, batch, size, slack=1.25, floor=4e-3):
  """Very deep nets (A-R152: 70 blocks) amplify bf16 rounding at random init beyond any fixed tolerance, so the
  bound is calibrated on the spot: at every named tap the product must be at least as close to the bf16-emulating
  oracle as that oracle is to its own fp32 evaluation (x slack).  A wiring or kernel bug shows up as an error well
  above the rounding noise at the first tap it touches."""
  from oracle import assembled_oracle as O
  om, pm = make_pair(name, device, batch, size)
  d = uses_d(name)
  _, x, _ = inputs(batch, size)
  lo = om(x, True, use_resnet_d=d).detach()
  taps_o = {k: v.detach().clone() for k, v in om.taps_nhwc().items()}
  lp = pm(x.to(device), True, use_resnet_d=d).float().cpu()
  of = O.Model(num_classes=1001, emulate_bf16=False, zero_gamma=True, seed=0, **CONFIGS[name])
  of(torch.zeros(2, size, size, 3), True, use_resnet_d=d)
  of.vars.pending_updates = {}
  with torch.no_grad():
    for n, t in om.vars.trainable.items():
      of.vars.trainable[n].copy_(t)
  lf = of(x, True, use_resnet_d=d).detach()
  taps_f = of.taps_nhwc()
  report = {}
  for k, v in taps_o.items():
    if k not in pm.taps:
      continue
    pv = lp if k == 'final_dense' else pm.taps[k].float().cpu().reshape(v.shape)
    e_p, e_f = util.rel_l2(pv, v), util.rel_l2(taps_f[k].detach(), v)
    report[k] = (e_p, e_f)
    assert e_p <= max(slack * e_f, floor), '%s: tap %s product-vs-oracle %.3e > %.2f x rounding noise %.3e' % (
        name, k, e_p, slack, e_f)
  assert util.rel_l2(lp, lo) <= max(slack * util.rel_l2(lf, lo), floor)
  return report


def check_backward(name, device, batch, size, label_smoothing=0.1, min_cos=0.8, min_global_cos=0.9):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  om, pm = make_pair(name, device, batch, size)
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


def check_train_steps(name, device, batch, size, steps, hp_kwargs, mixup_type=0, kd_temp=0.0, rel_tol=2e-2):
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
  assert 0.7 <= dec_p / dec_o <= 1.3, 'loss decrease %.4f vs oracle %.4f' % (dec_p, dec_o)
  # BN moving statistics were updated like the oracle's (UPDATE_OPS)
  some = [n for n in om.vars.state if n.endswith('moving_variance')][0]
  assert util.rel_l2(tr.model.arena.st(some).cpu(), om.vars.state[some]) <= 2e-2
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
