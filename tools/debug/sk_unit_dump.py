#!/usr/bin/env python
"""GPU debug helper: the fused SK unit's backward inputs vs the oracle's autograd values, per unit (teacher-forced run)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd import lib, ops  # noqa: E402
from tests import model_parity as mp  # noqa: E402

batch, size, kp = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ops.set_library(None, is_double=False)
lib.load()
cap = []
o_bwd = ops.sk_bn_bwd


def bwd(dv, att, ds, y, scale, shift, gamma, mean, invstd, dgamma, dbeta, F_, grad_stats=None, mask_stats=None):
  r = o_bwd(dv, att, ds, y, scale, shift, gamma, mean, invstd, dgamma, dbeta, F_, grad_stats, mask_stats)
  cap.append(dict(dv=dv.float().cpu(), att=att.float().cpu(), ds=ds.float().cpu(), y=y.float().cpu(), scale=scale.cpu().clone(),
                  shift=shift.cpu().clone(), dy=r.float().cpu(), F=F_, dbeta=dbeta.cpu().clone(), dgamma=dgamma.cpu().clone()))
  return r


ops.sk_bn_bwd = bwd
orc = {}
try:
  mp.check_teacher_forced_backward('a-r50-d', 'cuda', batch, size, keep_prob=kp, dx_tol=1, lazy_tol=1, dparam_tol=1, dw_tol=1,
                                   squeeze_tol=10, sk_tol=1, capture=orc)
finally:
  ops.sk_bn_bwd = o_bwd
units = [k[len('sk_out:'):] for k in orc['rec_extra'] if k.startswith('sk_out:')]
for k, g in enumerate(units):
  c = cap[len(cap) - 1 - k]
  F_ = c['F']
  fo = orc['rec_bn'][g][0]                       # oracle f = relu(bn(y)), NCHW, with .grad = df
  f_o = fo.detach().permute(0, 2, 3, 1)
  df_o = fo.grad.permute(0, 2, 3, 1)
  y = c['y']
  N, H, W, C2 = y.shape
  pre = y * c['scale'] + c['shift']
  f_p = torch.relu(pre).to(torch.bfloat16).float()
  mism = ((f_p > 0) != (f_o > 0))
  att = c['att'].view(N, 2 * F_)
  a0 = torch.softmax(torch.stack([att[:, :F_], att[:, F_:]], 0), 0)
  dv = c['dv']
  ds = c['ds'].view(N, 1, 1, F_) / (H * W)
  df_p = torch.cat([a0[0].view(N, 1, 1, F_) * dv + ds, a0[1].view(N, 1, 1, F_) * dv + ds], 3)
  dfm_p, dfm_o = df_p * (f_p > 0), df_o * (f_o > 0)
  pb = orc['om'].vars.trainable[g[:-5] + 'beta'].grad
  print('%2d %-52s HxW %3dx%-3d f rel %.1e mask mism %5d (%.2e) df rel %.2e masked-df rel %.2e dbeta rel %.2e  zero-input frac %.3f' % (
      k, g[-52:], H, W, float((f_p - f_o).norm() / f_o.norm()), int(mism.sum()), float(mism.float().mean()),
      float((df_p - df_o).norm() / df_o.norm()), float((dfm_p - dfm_o).norm() / dfm_o.norm()),
      float((c['dbeta'] - pb).norm() / pb.norm()), float((orc['rec_in'][g[:-len('batch_normalization_N/gamma')].rsplit('/', 1)[0] + '/x'] if False else torch.zeros(1)).mean())))
