"""Variable values as a pure function of (variable name, shape): both the fixture generator (which runs the REFERENCE
under the tf shim) and the tests (which run the ORACLE) fill their variables with these, so no weights are stored.
Non-trivial everywhere: block-final gammas are not zero, moving statistics are not (0, 1)."""
import math
import zlib

import numpy as np


def value_for(name: str, shape) -> np.ndarray:
  rng = np.random.default_rng(zlib.crc32(name.encode()) ^ 0x5eed)
  shape = tuple(int(s) for s in shape)
  leaf = name.rsplit('/', 1)[-1]
  if leaf == 'kernel' and len(shape) == 4:      # conv, HWIO
    fan_in = shape[0] * shape[1] * shape[2]
    return rng.standard_normal(shape) * math.sqrt(1.0 / fan_in)
  if leaf == 'kernel':                          # dense [in, out]
    lim = math.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape)
  if leaf == 'gamma':
    # damped (mean 0.3): with gammas around 1 a batch-2 training-mode pass is chaotic -- a 1e-15 perturbation of the
    # input grows to 1e-2 by block_layer4 of the 70-block Assemble-ResNet-152 (batch statistics over 2..8 samples
    # per channel), which would bury a float64 comparison; at this scale it stays below 1e-8
    return rng.uniform(0.15, 0.45, size=shape)
  if leaf in ('beta', 'bias', 'moving_mean'):
    return rng.normal(0.0, 0.1, size=shape)
  if leaf == 'moving_variance':
    return rng.uniform(0.5, 1.5, size=shape)
  raise KeyError('no rule for variable %s' % name)


def tap_summary(arr: np.ndarray, n_samples: int = 48):
  """compact, order-sensitive description of a tensor: shape, sum, sum of |.|, and samples at fixed flat indices"""
  flat = np.asarray(arr, dtype=np.float64).reshape(-1)
  idx = np.random.default_rng(flat.size).integers(0, flat.size, size=n_samples)
  return dict(shape=list(arr.shape), sum=float(flat.sum()), abs_sum=float(np.abs(flat).sum()), idx=idx.tolist(),
              vals=flat[idx].tolist())
