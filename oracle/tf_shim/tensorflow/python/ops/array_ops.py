import numpy as np
import torch

import tensorflow as _tf
from tensorflow import expand_dims, rank, reshape, shape, squeeze, stack, tile, transpose  # noqa: F401


def constant(value, dtype=None, shape=None, name=None):
  """[TF-sem] a Python float (list) without a dtype becomes a FLOAT32 constant: metric/ece_metric.py builds its bin edges
  this way (:216-221), so a float32 confidence equal to float32(0.1) lies ON the edge (bin 0), not above the double
  0.1.  The value is rounded to float32 and then carried in the shim's compute type."""
  if dtype is None and not isinstance(value, (_tf.Tensor, torch.Tensor)):
    arr = np.asarray(value)
    if arr.dtype.kind == 'f':
      return _tf.constant(arr.astype(np.float32).astype(np.float64), _tf.float32)
  return _tf.constant(value, dtype, shape, name)
