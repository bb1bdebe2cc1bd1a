"""Shared helpers for the parity tests (test infrastructure)."""
from __future__ import annotations

import numpy as np
import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
  a = a.detach().double().cpu().reshape(-1)
  b = b.detach().double().cpu().reshape(-1)
  den = float(b.norm())
  return float((a - b).norm()) / (den if den > 0 else 1.0)


def max_abs(a, b) -> float:
  return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def oracle_to_product_param(name: str, t: torch.Tensor) -> torch.Tensor:
  """oracle (TF layouts: conv HWIO, dense [in,out]) -> product (KRSC)."""
  if t.dim() == 4:
    return t.permute(3, 0, 1, 2).contiguous()
  if t.dim() == 2:
    return t.t().contiguous().view(t.shape[1], 1, 1, t.shape[0])
  return t


def product_to_oracle_grad(name: str, g: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
  if like.dim() == 4:
    return g.permute(1, 2, 3, 0)
  if like.dim() == 2:
    return g.view(g.shape[0], g.shape[3]).t()
  return g


def load_oracle_into_product(om, pm):
  """Copy every variable / moving statistic of the oracle model into the product model (same names)."""
  a = pm.arena
  assert list(om.vars.trainable.keys()) == list(a.specs.keys()), 'variable names / order differ'
  with torch.no_grad():
    for name, t in om.vars.trainable.items():
      a.w(name).copy_(oracle_to_product_param(name, t.detach().float()).to(a.w32.device))
    for name, t in om.vars.state.items():
      a.st(name).copy_(t.detach().float().to(a.w32.device))
  a.refresh_shadows()


def seeded_images(n, h, w, seed):
  rng = np.random.default_rng(seed)
  return torch.from_numpy(rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8))


def perturb_bn_state(om, seed):
  """Non-trivial moving statistics: mean ~ N(0, 0.1), var ~ U(0.5, 1.5)."""
  rng = np.random.default_rng(seed)
  for name in list(om.vars.state.keys()):
    t = om.vars.state[name]
    if name.endswith('moving_mean'):
      om.vars.state[name] = torch.from_numpy(rng.normal(0, 0.1, size=tuple(t.shape))).to(t.dtype)
    else:
      om.vars.state[name] = torch.from_numpy(rng.uniform(0.5, 1.5, size=tuple(t.shape))).to(t.dtype)


def set_knob(monkeypatch, name: str, value):
  """set an ASM_* variable for this test: the host-side switches are cached (ops.knob) and the kernel-selection ones live in
  the library's asm_tuning struct (it reads no environment), so both are refreshed"""
  from assembled_cnn_amd import ops
  monkeypatch.setenv(name, str(value))
  ops.refresh_tuning()
