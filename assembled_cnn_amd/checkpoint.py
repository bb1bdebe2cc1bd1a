"""Variable import / export in the reference's on-disk naming and layouts.

The reference stores TF-1 checkpoints whose variables are ``resnet_model/.../conv2d_N/kernel``
([k, k, Cin, Cout], HWIO), ``.../batch_normalization_N/{gamma,beta,moving_mean,moving_variance}``,
``.../sk_block*/sk_fc_{1,2}/kernel``, ``dense/{kernel,bias}`` ([in, out]) (nets/resnet_model.py:302-303).
This package keeps the same names and creation order but stores kernels KRSC; the functions below convert
both ways against a plain ``{name: numpy array}`` dictionary (what ``tf.train.load_checkpoint(...)`` /
``get_tensor`` yields -- TensorFlow itself is not needed, and not available, here) and implement the
warm-start rule of utils/hook_utils.py:29-56 (restore every trainable variable except non-SE ``dense``
layers, and only when global_step == 0).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from .model import Model

MOMENTUM_SLOT = '/Momentum'   # tf.train.MomentumOptimizer's slot variable name (nets/optimizer_setting.py:29)


def _is_dense_kernel(name: str) -> bool:
  """tf.layers.dense kernels ([in, units]) are the ones whose LAYER is named dense / dense_N; 'embedding_dense' is a
  tf.layers.conv2d (nets/resnet_model.py:576-580) and keeps the 4-D HWIO layout [1, 1, in, emb].  Also true for the
  kernel's optimiser slot ``<kernel>/Momentum``."""
  if name.endswith(MOMENTUM_SLOT):
    name = name[:-len(MOMENTUM_SLOT)]
  parts = name.split('/')
  if len(parts) < 2 or parts[-1] != 'kernel':
    return False
  layer = parts[-2]
  return layer == 'dense' or (layer.startswith('dense_') and layer[6:].isdigit())


def _to_tf_layout(name: str, t: torch.Tensor) -> np.ndarray:
  a = t.detach().float().cpu()
  if a.dim() == 4:
    if _is_dense_kernel(name):
      return a.view(a.shape[0], a.shape[3]).t().contiguous().numpy()      # [units,1,1,in] -> [in, units]
    return a.permute(1, 2, 3, 0).contiguous().numpy()                       # KRSC -> HWIO
  return a.numpy().copy()


def _from_tf_layout(name: str, a: np.ndarray, like: torch.Tensor) -> torch.Tensor:
  t = torch.as_tensor(np.asarray(a), dtype=torch.float32)
  if like.dim() == 4:
    if t.dim() == 2:                                                         # dense [in, units]
      if not _is_dense_kernel(name):
        raise ValueError('variable %s is a convolution kernel ([k, k, in, out] in the checkpoint), got a 2-D array' % name)
      t = t.t().contiguous().view(like.shape)
    else:
      t = t.permute(3, 0, 1, 2).contiguous()                                # HWIO -> KRSC
  if tuple(t.shape) != tuple(like.shape):
    raise ValueError('variable %s has shape %s in the checkpoint, the model expects %s (TF layout %s)'
                     % (name, tuple(np.asarray(a).shape), tuple(like.shape), 'HWIO' if like.dim() == 4 else 'as is'))
  return t



def export_variables(model: Model, global_step: Optional[int] = None, include_slots: bool = True
                     ) -> "OrderedDict[str, np.ndarray]":
  """All trainable variables, BN moving statistics and (``include_slots``) the momentum accumulators
  ``<variable>/Momentum`` an Estimator checkpoint of the reference holds, under their TF names, in TF layouts."""
  a = model.arena
  if not a.finalized:
    raise RuntimeError('build the model first')
  out: "OrderedDict[str, np.ndarray]" = OrderedDict()
  for name in a.specs:
    out[name] = _to_tf_layout(name, a.w(name))
  for name in a.state_specs:
    out[name] = a.st(name).detach().cpu().numpy().copy()
  if include_slots:
    for name in a.specs:
      out[name + MOMENTUM_SLOT] = _to_tf_layout(name, a.m(name))
  if global_step is not None:
    out['global_step'] = np.asarray(global_step, dtype=np.int64)
  return out


def warm_start_variable_names(model: Model) -> Iterable[str]:
  """utils/hook_utils.py:36-43: every trainable variable except 'dense' ones that are not in an se_block."""
  for name in model.arena.specs:
    if 'dense' in name and 'se_block' not in name:
      continue
    yield name


def import_variables(model: Model, variables: Dict[str, np.ndarray], warm_start: bool = False,
                     global_step: int = 0, strict: bool = True) -> Dict[str, list]:
  """Load a {TF name: array} dictionary.  ``warm_start`` applies the WarmStartHook rule (skip the classifier,
  only at global_step == 0, trainable variables only; momentum accumulators untouched).  A full restore also loads the
  ``<variable>/Momentum`` slots, zeroing those the checkpoint does not hold (``missing_slots``).  Returns the lists of
  loaded / skipped / missing names."""
  a = model.arena
  if not a.finalized:
    raise RuntimeError('build the model first (variables are created by a shape-only walk)')
  report = {'loaded': [], 'skipped': [], 'missing': [], 'missing_slots': []}
  if warm_start and global_step != 0:
    report['skipped'] = list(a.specs)
    return report
  wanted = set(warm_start_variable_names(model)) if warm_start else set(a.specs)
  with torch.no_grad():
    for name in a.specs:
      if name not in wanted:
        report['skipped'].append(name)
        continue
      if name not in variables:
        report['missing'].append(name)
        continue
      a.w(name).copy_(_from_tf_layout(name, variables[name], a.w(name)).to(a.w32.device))
      report['loaded'].append(name)
      slot = name + MOMENTUM_SLOT      # optional: inference / warm-start checkpoints carry no optimiser slots
      if not warm_start:
        if slot in variables:
          a.m(name).copy_(_from_tf_layout(name, variables[slot], a.m(name)).to(a.w32.device))
          report['loaded'].append(slot)
        else:
          # a full restore of a slot-less checkpoint must not keep whatever accumulator an earlier run left behind:
          # the slot starts at zero, as MomentumOptimizer creates it, and the report says so
          a.m(name).zero_()
          report['missing_slots'].append(slot)
    if not warm_start:   # a tf.train.Saver over trainables does not restore the moving statistics
      for name in a.state_specs:
        if name in variables:
          a.st(name).copy_(torch.as_tensor(np.asarray(variables[name]), dtype=torch.float32).to(a.w32.device))
          report['loaded'].append(name)
        else:
          report['missing'].append(name)
  if strict and report['missing']:
    raise KeyError('variables missing from the checkpoint: %s' % report['missing'][:5])
  a.refresh_shadows()
  return report


def save_npz(path: str, model: Model, global_step: Optional[int] = None):
  np.savez(path, **{k.replace('/', '|'): v for k, v in export_variables(model, global_step).items()})


def load_npz(path: str) -> Dict[str, np.ndarray]:
  with np.load(path) as z:
    return {k.replace('|', '/'): z[k] for k in z.files}


# ---- TensorFlow's own checkpoint files (tensor bundle) -----------------------------------------------------------------
def load_tf_checkpoint(path: str, names=None) -> Dict[str, np.ndarray]:
  """``path``: a checkpoint prefix (``.../model.ckpt-1234``) or a directory holding a ``checkpoint`` state file
  (tf.train.latest_checkpoint, utils/hook_utils.py:45-46) -> {TF name: array}, ready for import_variables."""
  import os
  from . import tf_bundle
  if os.path.isdir(path):
    latest = tf_bundle.latest_checkpoint(path)
    if latest is None:
      raise FileNotFoundError('no checkpoint state / index file under %s' % path)
    path = latest
  return tf_bundle.read_bundle(path, names)


def save_tf_checkpoint(prefix: str, model: Model, global_step: Optional[int] = None, include_slots: bool = True):
  """Write the model as a TensorFlow checkpoint (``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` + the
  ``checkpoint`` state file) under the reference's names and layouts -- loadable by the reference's tf.train.Saver."""
  import os
  from . import tf_bundle
  tf_bundle.write_bundle(prefix, export_variables(model, global_step, include_slots))
  tf_bundle.write_checkpoint_state(os.path.dirname(os.path.abspath(prefix)), os.path.basename(prefix))
