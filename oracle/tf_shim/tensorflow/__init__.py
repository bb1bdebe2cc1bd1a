"""A minimal, eager, torch-CPU-backed stand-in for the ``tensorflow`` 1.14 module -- TEST INFRASTRUCTURE ONLY.

Purpose: the reference's network code (nets/resnet_model.py, nets/blocks.py, nets/model_helper.py,
losses/cls_losses.py, utils/data_util.mixup, functions/model_fns.learning_rate_with_decay / keep_prob_decay) is
plain Python over ~50 ``tf.*`` calls.  TensorFlow 1.14 cannot be installed here, but with THIS module on sys.path as
``tensorflow`` those files import and run UNMODIFIED, which pins the oracle's *wiring* (scopes, variable names and
creation order, block / stride / shortcut selection, BigLittle merge, SK / SE plumbing, loss glue) against the
reference's own source (tests/golden/make_reference_taps.py generates fixtures, tests/test_reference_taps.py compares
the oracle with them).  What it can NOT pin is TensorFlow's behaviour inside an op: each op below restates the TF 1.14
rule it implements, in its general form (any stride / kernel / padding mode), marked [TF-sem]; none of it is derived
from the oracle.

Tensors are eager ``torch`` tensors (float64 by default, so wiring comparisons are exact to ~1e-12) carrying a
NOMINAL tf dtype; nothing here is imported by the product package.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from unittest import mock

import numpy as np
import torch
import torch.nn.functional as F

VERSION = __version__ = '1.14.0'
COMPUTE_DTYPE = torch.float64


# ---------------------------------------------------------------------------------------------------
# dtypes / shapes / tensors
# ---------------------------------------------------------------------------------------------------
class DType(object):
  def __init__(self, name, floating, torch_dtype):
    self.name, self.is_floating, self.is_integer, self._torch = name, floating, (not floating and name != 'bool'), torch_dtype

  def __eq__(self, other):
    return isinstance(other, DType) and other.name == self.name

  def __ne__(self, other):
    return not self.__eq__(other)

  def __hash__(self):
    return hash(self.name)

  def __repr__(self):
    return 'tf.' + self.name

  @property
  def base_dtype(self):
    return self


float16 = DType('float16', True, None)
float32 = DType('float32', True, None)
float64 = DType('float64', True, None)
bfloat16 = DType('bfloat16', True, None)
int32 = DType('int32', False, torch.int64)
int64 = DType('int64', False, torch.int64)
bool = DType('bool', False, torch.bool)   # noqa: A001  (tf.bool)


class TensorShape(object):
  def __init__(self, dims):
    self._d = [int(x) for x in dims]

  def as_list(self):
    return list(self._d)

  def assert_is_compatible_with(self, other):
    assert list(self._d) == list(other._d), 'shapes %s and %s are incompatible' % (self._d, other._d)

  @property
  def dims(self):
    return [_Dim(d) for d in self._d]

  def __getitem__(self, i):
    return self._d[i]

  def __len__(self):
    return len(self._d)

  def __iter__(self):
    return iter(self._d)

  @property
  def ndims(self):
    return len(self._d)

  def __repr__(self):
    return 'TensorShape(%s)' % self._d


class _Dim(object):
  def __init__(self, v):
    self.value = v

  def is_compatible_with(self, other):
    return self.value == int(other)


def set_compute_dtype(dt):
  """float64 (default: wiring comparisons exact to ~1e-12) or float32 (the arithmetic type of the reference's graph:
  needed where a float32 rounding decides an INTEGER, e.g. the resize target of _smallest_size_at_least)"""
  global COMPUTE_DTYPE
  COMPUTE_DTYPE = dt


def _nominal(t):
  if t.dtype == torch.bool:
    return bool
  if t.dtype.is_floating_point:
    return float32
  return int32


class Tensor(object):
  __array_priority__ = 1000

  def __init__(self, t, dtype=None, name=None):
    if isinstance(t, Tensor):
      t, dtype = t.t, (dtype or t.dtype)
    if t.dtype.is_floating_point and t.dtype != COMPUTE_DTYPE:
      t = t.to(COMPUTE_DTYPE)
    self.t = t
    self.dtype = dtype if dtype is not None else _nominal(t)
    self.name = name

  # ---- structure ----
  @property
  def shape(self):
    return TensorShape(self.t.shape)

  def get_shape(self):
    return self.shape

  def __len__(self):
    return self.t.shape[0]

  def __iter__(self):
    for i in range(self.t.shape[0]):
      yield Tensor(self.t[i], self.dtype)

  def __getitem__(self, idx):
    def plain(i):     # `logits[:tf.shape(x)[0]]`: slice bounds may be scalar tensors
      if isinstance(i, _builtins.slice):
        return _builtins.slice(*[(int(b.t.item()) if isinstance(b, Tensor) else b) for b in (i.start, i.stop, i.step)])
      return int(i.t.item()) if isinstance(i, Tensor) and i.t.dim() == 0 else i
    idx = tuple(plain(i) for i in idx) if isinstance(idx, tuple) else plain(idx)
    return Tensor(self.t[idx], self.dtype)

  def set_shape(self, shape):
    shape = list(shape)
    assert len(shape) == self.t.dim() and all(b is None or int(b) == int(a) for a, b in zip(self.t.shape, shape)), \
        'set_shape(%s) on a tensor of shape %s' % (shape, list(self.t.shape))

  def numpy(self):
    return self.t.detach().cpu().numpy()

  def __repr__(self):
    return 'tf.Tensor(shape=%s, dtype=%s)' % (list(self.t.shape), self.dtype.name)

  # ---- arithmetic (binary ops keep the left operand's nominal dtype, like TF's strict typing would require) ----
  def _bin(self, other, fn, reverse=False):
    o = _t(other)
    a, b = (o, self.t) if reverse else (self.t, o)
    r = fn(a, b)
    return Tensor(r, self.dtype if r.dtype != torch.bool else bool)

  def __add__(self, o): return self._bin(o, torch.add)
  def __radd__(self, o): return self._bin(o, torch.add, True)
  def __sub__(self, o): return self._bin(o, torch.sub)
  def __rsub__(self, o): return self._bin(o, torch.sub, True)
  def __mul__(self, o): return self._bin(o, torch.mul)
  def __rmul__(self, o): return self._bin(o, torch.mul, True)
  def __truediv__(self, o): return self._bin(o, torch.true_divide)
  def __rtruediv__(self, o): return self._bin(o, torch.true_divide, True)
  def __floordiv__(self, o): return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode='floor'))
  def __neg__(self): return Tensor(-self.t, self.dtype)
  def __lt__(self, o): return self._bin(o, torch.lt)
  def __le__(self, o): return self._bin(o, torch.le)
  def __gt__(self, o): return self._bin(o, torch.gt)
  def __ge__(self, o): return self._bin(o, torch.ge)
  def __pow__(self, o): return self._bin(o, torch.pow)

  def __bool__(self):
    return builtins_bool(self.t.item())

  def __int__(self):
    return int(self.t.item())

  def __float__(self):
    return float(self.t.item())
  # no __index__: `list_of_tensors * tensor` (nets/blocks.py:152) must reach Tensor.__rmul__, not list repetition


import builtins as _builtins  # noqa: E402
builtins_bool = _builtins.bool


def _t(x):
  """anything -> torch tensor (lists of tensors are stacked, as tf.convert_to_tensor packs them)"""
  if isinstance(x, Tensor):
    return x.t
  if isinstance(x, torch.Tensor):
    return x if not x.dtype.is_floating_point else x.to(COMPUTE_DTYPE)
  if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], (Tensor, torch.Tensor)):
    return torch.stack([_t(e) for e in x], 0)
  if isinstance(x, np.ndarray):
    return _t(torch.from_numpy(np.ascontiguousarray(x)))
  if isinstance(x, (list, tuple)):
    return _t(np.asarray(x))
  if isinstance(x, float):
    return torch.tensor(x, dtype=COMPUTE_DTYPE)
  if isinstance(x, builtins_bool):
    return torch.tensor(x)
  if isinstance(x, int):
    return torch.tensor(x, dtype=torch.int64)
  if isinstance(x, np.generic):
    return _t(x.item())
  raise TypeError('tf shim: cannot convert %r' % type(x))


def _wrap(t, like=None, dtype=None):
  if dtype is None and isinstance(like, Tensor):
    dtype = like.dtype if t.dtype.is_floating_point == like.t.dtype.is_floating_point else None
  return Tensor(t, dtype)


def _ints(x):
  if isinstance(x, Tensor):
    return [int(v) for v in x.t.reshape(-1).tolist()]
  if isinstance(x, (int, np.integer)):
    return int(x)
  return [int(v) if not isinstance(v, Tensor) else int(v.t.item()) for v in x]


# ---------------------------------------------------------------------------------------------------
# graph state: variables, scopes, named taps (tf.identity(x, name)), update ops, logging
# ---------------------------------------------------------------------------------------------------
class _Graph(object):
  def __init__(self):
    self.variables = OrderedDict()      # full name -> Variable, creation order == tf.global_variables()
    self.scope = []                     # current variable-scope path components (already uniquified)
    self.scope_counts = {}              # VariableScopeStore.variable_scopes_count [TF-sem]
    self.reuse = False
    self.named = OrderedDict()          # tf.identity(x, name) taps, last write wins
    self.update_ops = []                # tf.GraphKeys.UPDATE_OPS: (variable, new value)
    self.rng = np.random.default_rng(0)
    self.uniform_draws = []             # every tf.random_uniform draw, in order (DropBlock)
    self.beta_draws = []                # every Beta.sample draw (mixup)
    self.track_grad = False             # trainable variables become autograd leaves (optimizer.compute_gradients)
    self.opt_slots = OrderedDict()      # '<variable>/Momentum' -> accumulator (tf.train.MomentumOptimizer slots)
    self.last_grads = OrderedDict()     # variable name -> gradient handed to apply_gradients (after any rescaling)
    self.metric_vars = OrderedDict()    # local (metric) variables by full name
    self.window_draws = []              # every sample_distorted_bounding_box / random_flip draw (input pipeline)


_G = _Graph()


def reset_default_graph():
  global _G
  _G = _Graph()


def get_default_graph():
  return _G


def shim_state():
  return _G


class Variable(Tensor):
  def __init__(self, t, name, trainable):
    Tensor.__init__(self, t, float32, name)
    self.trainable = trainable

  def assign(self, value):
    t = _t(value)
    self.t = (t.to(COMPUTE_DTYPE) if t.dtype.is_floating_point else t).detach().clone()
    if _G.track_grad and self.trainable:
      self.t.requires_grad_(True)
    return self


def trainable_variables():
  return [v for v in _G.variables.values() if v.trainable]


def global_variables():
  return list(_G.variables.values())


class variable_scope(object):
  """tf.variable_scope(name_or_scope, default_name=None, values=None, reuse=None, custom_getter=None).

  [TF-sem] (tensorflow/python/ops/variable_scope.py, 1.14): with name_or_scope None the scope is
  ``default_name`` made unique among the variable scopes OPENED so far under the current one (name, name_1, ...,
  ``_get_unique_variable_scope``: first idx whose ``variable_scope_count(current/name_idx)`` is 0); entering a scope
  increments its count; leaving a scope resets the counts of all of ITS sub-scopes (``close_variable_subscopes``),
  which is what makes a second pass under reuse=True regenerate the same names.  A custom_getter is honoured only in
  so far as the reference's own getter is the identity for float32 (nets/resnet_model.py:251-290)."""

  def __init__(self, name_or_scope, default_name=None, values=None, reuse=None, custom_getter=None, **_):
    self.name, self.default_name, self.reuse = name_or_scope, default_name, reuse

  def _full(self, name):
    return '/'.join(_G.scope + [name])

  def __enter__(self):
    name = self.name
    if isinstance(name, variable_scope):      # `with tf.variable_scope(...) as name:` re-entry
      name = name._entered
    if name is None:
      base, idx, name = self.default_name, 0, self.default_name
      while _G.scope_counts.get(self._full(name), 0) > 0:
        idx += 1
        name = '%s_%d' % (base, idx)
    full = self._full(name)
    _G.scope_counts[full] = _G.scope_counts.get(full, 0) + 1
    _G.scope.append(name)
    self._entered = name
    self._saved_reuse = _G.reuse
    if self.reuse:
      _G.reuse = True
    return self

  def __exit__(self, *exc):
    full = '/'.join(_G.scope)
    for k in list(_G.scope_counts.keys()):
      if k.startswith(full + '/'):
        _G.scope_counts[k] = 0
    _G.scope.pop()
    _G.reuse = self._saved_reuse
    return False


class name_scope(object):
  def __init__(self, name, default_name=None, values=None):
    self.name = name or default_name

  def __enter__(self):
    return self.name

  def __exit__(self, *exc):
    return False


def _var_rng(full_name):
  return np.random.default_rng(zlib.crc32(full_name.encode()))


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **_):
  full = '/'.join(_G.scope + [name])
  if full in _G.variables:
    if not _G.reuse:
      raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True?' % full)
    return _G.variables[full]
  if _G.reuse:
    raise ValueError('Variable %s does not exist, or was not created with tf.get_variable()' % full)
  shape = [int(s) for s in shape]
  if initializer is None:
    initializer = glorot_uniform_initializer()
  val = initializer(shape, _var_rng(full)) if callable(initializer) else np.broadcast_to(np.asarray(initializer), shape)
  v = Variable(torch.from_numpy(np.ascontiguousarray(np.asarray(val, dtype=np.float64))), full, trainable)
  if _G.track_grad and trainable:
    v.t.requires_grad_(True)
  _G.variables[full] = v
  return v


# initialisers [TF-sem]: distributions as in TF 1.14; the STREAM is this shim's own (seeded by the variable name)
class zeros_initializer(object):
  def __call__(self, shape, rng=None):
    return np.zeros(shape)


class ones_initializer(object):
  def __call__(self, shape, rng=None):
    return np.ones(shape)


class constant_initializer(object):
  def __init__(self, value=0):
    self.value = value

  def __call__(self, shape, rng=None):
    return np.full(shape, float(self.value))


def _fans(shape):
  if len(shape) == 2:
    return shape[0], shape[1]
  rf = int(np.prod(shape[:-2]))
  return shape[-2] * rf, shape[-1] * rf


class variance_scaling_initializer(object):
  """tf.variance_scaling_initializer(scale=1.0, mode='fan_in', distribution='truncated_normal'):
  stddev = sqrt(scale / fan_in) / .87962566103423978, resampled into +-2 stddev."""

  def __init__(self, scale=1.0, mode='fan_in', distribution='truncated_normal'):
    self.scale, self.mode, self.distribution = scale, mode, distribution

  def __call__(self, shape, rng):
    fan_in, fan_out = _fans(shape)
    n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.0}[self.mode]
    std = math.sqrt(self.scale / max(1.0, n)) / .87962566103423978
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
      out[bad] = rng.standard_normal(size=int(bad.sum()))
      bad = np.abs(out) > 2.0
    return out * std


class glorot_uniform_initializer(object):
  def __call__(self, shape, rng):
    fan_in, fan_out = _fans(shape)
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


# ---------------------------------------------------------------------------------------------------
# basic ops
# ---------------------------------------------------------------------------------------------------
def constant(value, dtype=None, shape=None, name=None):
  t = _t(value)
  if dtype is not None and dtype.is_floating:
    t = t.to(COMPUTE_DTYPE)
  return Tensor(t, dtype if dtype is not None else None)


def convert_to_tensor(value, dtype=None, name=None):
  return constant(value, dtype)


def identity(x, name=None):
  out = Tensor(_t(x), getattr(x, 'dtype', None))
  if name is not None:
    _G.named[name] = out
  return out


def stop_gradient(x, name=None):
  return Tensor(_t(x).detach(), getattr(x, 'dtype', None))


def cast(x, dtype, name=None):
  t = _t(x)
  if dtype.is_floating:
    return Tensor(t.to(COMPUTE_DTYPE), dtype)
  return Tensor(t.to(dtype._torch), dtype)


def to_float(x, name=None):
  return cast(x, float32)


def shape(x, name=None):
  return Tensor(torch.tensor(list(_t(x).shape), dtype=torch.int64), int32)


def size(x, name=None):
  return Tensor(torch.tensor(_t(x).numel(), dtype=torch.int64), int32)


def reshape(x, shp, name=None):
  return _wrap(_t(x).reshape(_ints(shp)), x)


def transpose(x, perm=None, name=None):
  t = _t(x)
  if perm is None:     # [TF-sem] default perm reverses the dimensions
    return _wrap(t.permute(*reversed(range(t.dim()))), x)
  return _wrap(t.permute(*_ints(perm)), x)


def expand_dims(x, axis=None, name=None, dim=None):
  return _wrap(_t(x).unsqueeze(axis if axis is not None else dim), x)


def squeeze(x, axis=None, name=None, squeeze_dims=None):
  axis = axis if axis is not None else squeeze_dims
  t = _t(x)
  if axis is None:
    return _wrap(t.squeeze(), x)
  for a in sorted(_ints(axis) if not isinstance(axis, int) else [axis], reverse=True):
    assert t.shape[a] == 1, 'tf.squeeze of a non-unit dimension'
    t = t.squeeze(a)
  return _wrap(t, x)


def tile(x, multiples, name=None):
  return _wrap(_t(x).repeat(*_ints(multiples)), x)


def stack(values, axis=0, name=None):
  if all(isinstance(v, (int, np.integer)) for v in values):
    return Tensor(torch.tensor([int(v) for v in values], dtype=torch.int64), int32)
  return Tensor(torch.stack([_t(v) for v in values], axis))


def concat(values, axis, name=None):
  return _wrap(torch.cat([_t(v) for v in values], axis), values[0])


def split(value, num_or_size_splits, axis=0, num=None, name=None):
  t = _t(value)
  if isinstance(num_or_size_splits, (int, np.integer)):
    n = int(num_or_size_splits)
    assert t.shape[axis] % n == 0, 'tf.split: dimension not divisible'
    parts = torch.split(t, t.shape[axis] // n, dim=axis)
  else:
    parts = torch.split(t, _ints(num_or_size_splits), dim=axis)
  return [_wrap(p, value) for p in parts]


def reverse(x, axis, name=None):
  return _wrap(torch.flip(_t(x), _ints(axis)), x)


def pad(x, paddings, mode='CONSTANT', name=None, constant_values=0):
  """tf.pad: paddings[d] = [before, after] per dimension; REFLECT mirrors without repeating the edge [TF-sem]."""
  t = _t(x)
  p = [[int(a), int(b)] for a, b in paddings]
  assert len(p) == t.dim()
  if mode.upper() == 'CONSTANT':
    flat = []
    for a, b in reversed(p):
      flat += [a, b]
    return _wrap(F.pad(t, flat, value=constant_values), x)
  if mode.upper() == 'REFLECT':
    for d, (a, b) in enumerate(p):
      if a == 0 and b == 0:
        continue
      n = t.shape[d]
      assert a < n and b < n, 'REFLECT padding must be smaller than the dimension'
      idx = list(range(a, 0, -1)) + list(range(n)) + list(range(n - 2, n - 2 - b, -1))
      t = t.index_select(d, torch.tensor(idx))
    return _wrap(t, x)
  raise NotImplementedError(mode)


def _axes(axis):
  if axis is None:
    return None
  return _ints(axis) if not isinstance(axis, (int, np.integer)) else [int(axis)]


def _reduce(fn, x, axis, keepdims, keep_dims):
  kd = builtins_bool(keepdims) or builtins_bool(keep_dims)
  t = _t(x)
  ax = _axes(axis)
  if ax is None:
    ax = list(range(t.dim()))
  like = x if isinstance(x, Tensor) else (x[0] if isinstance(x, (list, tuple)) and isinstance(x[0], Tensor) else None)
  return _wrap(fn(t, dim=ax, keepdim=kd), like)


def reduce_sum(x, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
  return _reduce(torch.sum, x, axis if axis is not None else reduction_indices, keepdims, keep_dims)


def reduce_mean(x, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
  return _reduce(torch.mean, x, axis if axis is not None else reduction_indices, keepdims, keep_dims)


def multiply(a, b, name=None):
  if isinstance(a, Tensor):
    return a * b
  if isinstance(b, Tensor):
    return b * a
  return Tensor(_t(a) * _t(b))


def add(a, b, name=None):
  return Tensor(_t(a) + _t(b), getattr(a, 'dtype', None))


def maximum(a, b, name=None):
  return _wrap(torch.maximum(_t(a), _t(b).to(_t(a).dtype)), a if isinstance(a, Tensor) else b)


def minimum(a, b, name=None):
  return _wrap(torch.minimum(_t(a), _t(b).to(_t(a).dtype)), a if isinstance(a, Tensor) else b)


def pow(x, y, name=None):  # noqa: A001
  tx, ty = _t(x), _t(y)
  if not tx.dtype.is_floating_point:
    tx = tx.to(COMPUTE_DTYPE)
  return _wrap(torch.pow(tx, ty), x if isinstance(x, Tensor) else None)


def sign(x, name=None):
  return _wrap(torch.sign(_t(x)), x)


def clip_by_value(x, lo, hi, name=None):
  return _wrap(torch.clamp(_t(x), min=float(lo), max=float(hi)), x)


def cond(pred, true_fn=None, false_fn=None, **_):
  return true_fn() if builtins_bool(_t(pred).item()) else false_fn()


def group(*a, **k):
  """tf.group(minimize_op, update_ops): the shim is eager, so the minimize op has already run; what is left to run are
  the pending (variable, value) pairs of UPDATE_OPS handed in (nets/optimizer_setting.py:36-37)"""
  for item in a:
    if isinstance(item, (list, tuple)):
      for op in item:
        if isinstance(op, tuple) and len(op) == 2 and isinstance(op[0], Variable):
          op[0].assign(op[1])
          _G.update_ops = [u for u in _G.update_ops if u is not op]
  return None


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):  # noqa: A002
  maxval = 1.0 if maxval is None else maxval
  u = _G.rng.uniform(minval, maxval, size=_ints(shape))
  t = torch.from_numpy(u)
  _G.uniform_draws.append(t.clone())
  return Tensor(t, dtype)


# ---------------------------------------------------------------------------------------------------
# tf.nn / tf.layers (channels_last; the reference picks NHWC when tf.test.is_built_with_cuda() is False)
# ---------------------------------------------------------------------------------------------------
def _same_pads(in_size, k, s):
  """[TF-sem] padding='SAME': out = ceil(in / s); total = max((out - 1) * s + k - in, 0); before = total // 2."""
  out = -(-in_size // s)
  total = max((out - 1) * s + k - in_size, 0)
  return total // 2, total - total // 2


def _nhwc_to_nchw(t):
  return t.permute(0, 3, 1, 2)


def _nchw_to_nhwc(t):
  return t.permute(0, 2, 3, 1)


def _conv_nhwc(t, filt_hwio, strides_hw, padding):
  """cross-correlation, filter [kh, kw, in/groups, out]; groups = C_in / filter_in (TF 1.14 GPU grouped conv)"""
  kh, kw, fin, fout = filt_hwio.shape
  cin = t.shape[3]
  assert cin % fin == 0
  groups = cin // fin
  x = _nhwc_to_nchw(t)
  if padding.upper() == 'SAME':
    ph = _same_pads(t.shape[1], kh, strides_hw[0])
    pw = _same_pads(t.shape[2], kw, strides_hw[1])
    x = F.pad(x, [pw[0], pw[1], ph[0], ph[1]])
  else:
    assert padding.upper() == 'VALID'
  w = filt_hwio.permute(3, 2, 0, 1)
  return _nchw_to_nhwc(F.conv2d(x, w, stride=tuple(strides_hw), groups=groups))


def _pair(v):
  return (int(v), int(v)) if isinstance(v, (int, np.integer)) else (int(v[0]), int(v[1]))


class _NN(object):
  @staticmethod
  def relu(x, name=None):
    return _wrap(torch.relu(_t(x)), x)

  @staticmethod
  def sigmoid(x, name=None):
    return _wrap(torch.sigmoid(_t(x)), x)

  @staticmethod
  def softmax(logits, axis=-1, name=None, dim=None):
    ax = axis if dim is None else dim
    like = logits if isinstance(logits, Tensor) else logits[0]
    return _wrap(torch.softmax(_t(logits), dim=ax), like)

  @staticmethod
  def log_softmax(logits, axis=-1, name=None):
    return _wrap(torch.log_softmax(_t(logits), dim=axis), logits)

  @staticmethod
  def conv2d(input, filter=None, strides=None, padding=None, data_format='NHWC', name=None, filters=None, **_):  # noqa: A002
    assert data_format == 'NHWC', 'the shim runs channels_last'
    f = _t(filter if filter is not None else filters)
    s = _ints(strides)
    return _wrap(_conv_nhwc(_t(input), f, (s[1], s[2]), padding), input)

  @staticmethod
  def sigmoid_cross_entropy_with_logits(labels=None, logits=None, name=None, _sentinel=None):
    """[TF-sem] max(x, 0) - x * z + log(1 + exp(-|x|))"""
    x, z = _t(logits), _t(labels)
    return _wrap(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs())), logits)

  @staticmethod
  def l2_loss(t, name=None):
    return Tensor((_t(t) ** 2).sum() / 2)

  @staticmethod
  def in_top_k(predictions, targets, k, name=None):
    p, tg = _t(predictions), _t(targets)
    tv = p.gather(1, tg.view(-1, 1))
    return Tensor((p > tv).sum(1) < k)


nn = _NN()


def _layer_vars(default_name, name):
  return variable_scope(name, default_name=default_name)


class _Layers(object):
  @staticmethod
  def conv2d(inputs, filters, kernel_size, strides=(1, 1), padding='valid', data_format='channels_last',
             use_bias=True, kernel_initializer=None, bias_initializer=None, name=None, **_):
    assert data_format == 'channels_last'
    k, s = _pair(kernel_size), _pair(strides)
    cin = _t(inputs).shape[3]
    with _layer_vars('conv2d', name):
      kernel = get_variable('kernel', [k[0], k[1], cin, int(filters)], float32,
                            kernel_initializer or glorot_uniform_initializer())
      bias = get_variable('bias', [int(filters)], float32, bias_initializer or zeros_initializer()) if use_bias else None
    y = _conv_nhwc(_t(inputs), kernel.t, s, padding)
    if bias is not None:
      y = y + bias.t
    return _wrap(y, inputs)

  @staticmethod
  def dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
            name=None, **_):
    cin = _t(inputs).shape[-1]
    with _layer_vars('dense', name):
      kernel = get_variable('kernel', [cin, int(units)], float32, kernel_initializer or glorot_uniform_initializer())
      bias = get_variable('bias', [int(units)], float32, bias_initializer or zeros_initializer()) if use_bias else None
    y = _t(inputs) @ kernel.t
    if bias is not None:
      y = y + bias.t
    return _wrap(y, inputs)

  @staticmethod
  def batch_normalization(inputs, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, training=False,
                          fused=None, gamma_initializer=None, beta_initializer=None, name=None, **_):
    """tf.layers.batch_normalization(fused=True) [TF-sem, 1.14]: training -> normalise with the batch mean and the
    BIASED batch variance over every axis but `axis`; the moving variance is updated with the Bessel-corrected
    (n / (n - 1)) batch variance (the fused kernel's "reserve" output), moving = moving * momentum + batch * (1 -
    momentum), registered in UPDATE_OPS (not applied until the train op runs).  Inference -> moving statistics."""
    t = _t(inputs)
    ax = axis if axis >= 0 else t.dim() + axis
    c = t.shape[ax]
    with _layer_vars('batch_normalization', name):
      gamma = get_variable('gamma', [c], float32, gamma_initializer or ones_initializer()) if scale else None
      beta = get_variable('beta', [c], float32, beta_initializer or zeros_initializer()) if center else None
      mm = get_variable('moving_mean', [c], float32, zeros_initializer(), trainable=False)
      mv = get_variable('moving_variance', [c], float32, ones_initializer(), trainable=False)
    red = [d for d in range(t.dim()) if d != ax]
    bshape = [1] * t.dim()
    bshape[ax] = c
    if training:
      mean = t.mean(dim=red)
      var = ((t - mean.view(bshape)) ** 2).mean(dim=red)
      n = t.numel() // c
      unbiased = var * (float(n) / float(max(n - 1, 1)))
      _G.update_ops.append((mm, mm.t * momentum + mean * (1.0 - momentum)))
      _G.update_ops.append((mv, mv.t * momentum + unbiased * (1.0 - momentum)))
    else:
      mean, var = mm.t, mv.t
    y = (t - mean.view(bshape)) * torch.rsqrt(var + epsilon).view(bshape)
    if gamma is not None:
      y = y * gamma.t.view(bshape)
    if beta is not None:
      y = y + beta.t.view(bshape)
    return _wrap(y, inputs)

  @staticmethod
  def _pool(inputs, pool_size, strides, padding, data_format, kind):
    assert data_format == 'channels_last'
    k, s = _pair(pool_size), _pair(strides)
    t = _t(inputs)
    x = _nhwc_to_nchw(t)
    if padding.upper() == 'SAME':
      ph, pw = _same_pads(t.shape[1], k[0], s[0]), _same_pads(t.shape[2], k[1], s[1])
    else:
      ph = pw = (0, 0)
    if kind == 'max':   # [TF-sem] padded cells never win
      x = F.pad(x, [pw[0], pw[1], ph[0], ph[1]], value=float('-inf'))
      y = F.max_pool2d(x, k, s)
    else:               # [TF-sem] SAME average pooling divides by the number of IN-RANGE cells
      ones = torch.ones_like(x[:1, :1])
      xs = F.avg_pool2d(F.pad(x, [pw[0], pw[1], ph[0], ph[1]]), k, s, divisor_override=1)
      cnt = F.avg_pool2d(F.pad(ones, [pw[0], pw[1], ph[0], ph[1]]), k, s, divisor_override=1)
      y = xs / cnt
    return _wrap(_nchw_to_nhwc(y), inputs)

  @staticmethod
  def max_pooling2d(inputs, pool_size, strides, padding='valid', data_format='channels_last', name=None):
    return _Layers._pool(inputs, pool_size, strides, padding, data_format, 'max')

  @staticmethod
  def average_pooling2d(inputs, pool_size, strides, padding='valid', data_format='channels_last', name=None):
    return _Layers._pool(inputs, pool_size, strides, padding, data_format, 'avg')

  @staticmethod
  def flatten(inputs, name=None, data_format='channels_last'):
    t = _t(inputs)
    return _wrap(t.reshape(t.shape[0], -1), inputs)


layers = _Layers()


class _UpSampling2D(object):
  def __init__(self, size=(2, 2), data_format=None, **_):
    assert data_format in (None, 'channels_last')
    self.size = _pair(size)

  def __call__(self, x):
    t = _t(x)   # nearest neighbour: every pixel repeated size[0] x size[1] [TF-sem]
    return _wrap(t.repeat_interleave(self.size[0], dim=1).repeat_interleave(self.size[1], dim=2), x)


class _Losses(object):
  @staticmethod
  def softmax_cross_entropy(onehot_labels, logits, weights=1.0, label_smoothing=0, scope=None, **_):
    """[TF-sem] tf.losses.softmax_cross_entropy: targets = onehot * (1 - ls) + ls / num_classes; per-row CE;
    Reduction.SUM_BY_NONZERO_WEIGHTS with a scalar weight = weight * mean over the batch."""
    z, y = _t(logits), _t(onehot_labels).to(COMPUTE_DTYPE)
    if label_smoothing > 0:
      nc = y.shape[1]
      y = y * (1.0 - label_smoothing) + label_smoothing / nc
    rows = -(y * torch.log_softmax(z, dim=1)).sum(1)
    w = _t(weights).to(COMPUTE_DTYPE)
    if w.dim() == 0:
      return Tensor(rows.sum() * w / (rows.numel() if float(w) != 0.0 else 1.0))
    num = (w != 0).sum().clamp(min=1)
    return Tensor((rows * w).sum() / num)


losses = _Losses()


class _Beta(object):
  def __init__(self, a, b):
    self.a, self.b = float(a), float(b)

  def sample(self, shape):
    n = _ints(shape)
    d = torch.from_numpy(_G.rng.beta(self.a, self.b, size=n))
    _G.beta_draws.append(d.clone())
    return Tensor(d, float32)


class _Train(object):
  """learning-rate schedules of tf.train [TF-sem, 1.14] evaluated eagerly on an integer global_step"""
  @staticmethod
  def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
    p = float(_t(global_step)) / float(decay_steps)
    if staircase:
      p = math.floor(p)
    return Tensor(torch.tensor(learning_rate * decay_rate ** p, dtype=COMPUTE_DTYPE))

  @staticmethod
  def polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False,
                       name=None):
    assert not cycle
    g = min(float(_t(global_step)), float(decay_steps))
    return Tensor(torch.tensor((learning_rate - end_learning_rate) * (1 - g / float(decay_steps)) ** power +
                               end_learning_rate, dtype=COMPUTE_DTYPE))

  @staticmethod
  def piecewise_constant(x, boundaries, values, name=None):
    """values[0] for x <= boundaries[0], values[i] for boundaries[i-1] < x <= boundaries[i], values[-1] beyond"""
    g = float(_t(x))
    for b, v in zip(boundaries, values):
      if g <= b:
        return Tensor(torch.tensor(float(v), dtype=COMPUTE_DTYPE))
    return Tensor(torch.tensor(float(values[-1]), dtype=COMPUTE_DTYPE))

  @staticmethod
  def cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0, name=None):
    g = min(float(_t(global_step)), float(decay_steps))
    cd = 0.5 * (1 + math.cos(math.pi * g / float(decay_steps)))
    return Tensor(torch.tensor(learning_rate * ((1 - alpha) * cd + alpha), dtype=COMPUTE_DTYPE))

  def __getattr__(self, name):
    return mock.MagicMock(name='tf.train.' + name)


train = _Train()


class _Namespace(object):
  def __init__(self, **kw):
    self.__dict__.update(kw)

  def __getattr__(self, name):
    return mock.MagicMock(name=name)


keras = _Namespace(layers=_Namespace(UpSampling2D=_UpSampling2D))
contrib = _Namespace(distributions=_Namespace(Beta=_Beta))
distributions = _Namespace(Beta=_Beta)
test = _Namespace(is_built_with_cuda=lambda: False)   # -> the reference picks data_format='channels_last'
logging = _Namespace(info=lambda *a, **k: None, warn=lambda *a, **k: None, warning=lambda *a, **k: None,
                     debug=lambda *a, **k: None, error=lambda *a, **k: None, INFO=20, set_verbosity=lambda *a: None)


class _GraphKeys(object):
  UPDATE_OPS = 'update_ops'
  TRAINABLE_VARIABLES = 'trainable_variables'
  GLOBAL_VARIABLES = 'variables'


GraphKeys = _GraphKeys


def get_collection(key, scope=None):
  if key == _GraphKeys.UPDATE_OPS:
    return list(_G.update_ops)
  if key == _GraphKeys.TRAINABLE_VARIABLES:
    return trainable_variables()
  return global_variables()


def apply_update_ops():
  """run what tf.group(minimize_op, update_ops) would (nets/optimizer_setting.py:36-37)"""
  for var, val in _G.update_ops:
    var.assign(val)
  _G.update_ops = []


# ---------------------------------------------------------------------------------------------------
# round 3: what resnet_model_fn / get_train_op / preprocess_image / metric.ece_metric need beyond the network code
# ---------------------------------------------------------------------------------------------------
def argmax(x, axis=None, name=None, dimension=None, output_type=None):
  ax = axis if axis is not None else (dimension if dimension is not None else 0)
  return Tensor(torch.argmax(_t(x), dim=int(ax)), int64)   # [TF-sem] first maximal index on ties (torch agrees)


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
  assert on_value is None and off_value is None and axis in (None, -1)
  return Tensor(F.one_hot(_t(indices).long(), int(depth)).to(COMPUTE_DTYPE), dtype or float32)


def add_n(inputs, name=None):
  out = _t(inputs[0])
  for v in inputs[1:]:
    out = out + _t(v)
  return _wrap(out, inputs[0] if isinstance(inputs[0], Tensor) else None)


def reduce_max(x, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
  return _reduce(torch.amax, x, axis if axis is not None else reduction_indices, keepdims, keep_dims)


def greater(a, b, name=None):
  return Tensor(torch.gt(_t(a), _t(b)), bool)


def less_equal(a, b, name=None):
  return Tensor(torch.le(_t(a), _t(b)), bool)


def equal(a, b, name=None):
  return Tensor(torch.eq(_t(a), _t(b)), bool)


def logical_and(a, b, name=None):
  return Tensor(torch.logical_and(_t(a), _t(b)), bool)


def abs(x, name=None):  # noqa: A001
  return _wrap(torch.abs(_t(x)), x)


def div(a, b, name=None):
  ta, tb = _t(a), _t(b)
  assert ta.dtype.is_floating_point or tb.dtype.is_floating_point, 'tf.div on integers floors; not needed here'
  return _wrap(ta / tb, a if isinstance(a, Tensor) else b)


def rank(x, name=None):
  return Tensor(torch.tensor(_t(x).dim(), dtype=torch.int64), int32)


def broadcast_to(x, shp, name=None):
  t = _t(x)
  if not t.dtype.is_floating_point:
    t = t.to(COMPUTE_DTYPE)
  return Tensor(torch.broadcast_to(t.to(COMPUTE_DTYPE), _ints(shp)).clone(), float32)


def slice(x, begin, size, name=None):  # noqa: A001
  """tf.slice: size -1 = to the end of the dimension"""
  t = _t(x)
  b, n = _ints(begin), _ints(size)
  idx = tuple(_builtins.slice(bi, (t.shape[d] if ni == -1 else bi + ni)) for d, (bi, ni) in enumerate(zip(b, n)))
  return _wrap(t[idx], x)


def unstack(x, num=None, axis=0, name=None):
  t = _t(x)
  return [_wrap(e, x) for e in torch.unbind(t, dim=axis)]


def zeros(shp, dtype=float32, name=None):
  if dtype.is_floating:
    return Tensor(torch.zeros(_ints(shp), dtype=COMPUTE_DTYPE), dtype)
  return Tensor(torch.zeros(_ints(shp), dtype=dtype._torch), dtype)


def to_int32(x, name=None):
  return cast(x, int32)    # [TF-sem] float -> int casts truncate toward zero (torch agrees)


_expand_dims_scalar = expand_dims


def expand_dims(x, axis=None, name=None, dim=None):  # noqa: F811  (array_ops.expand_dims(x, [1]) passes a list)
  a = axis if axis is not None else dim
  if isinstance(a, (list, tuple)):
    assert len(a) == 1
    a = a[0]
  return _expand_dims_scalar(x, a)


# ---- tf.train: global step + MomentumOptimizer -----------------------------------------------------------------
def _get_or_create_global_step(graph=None):
  gs = _G.variables.get('global_step')
  if gs is None:
    gs = Variable(torch.zeros((), dtype=torch.int64), 'global_step', False)
    gs.dtype = int64
    _G.variables['global_step'] = gs
  return gs


class _MomentumOptimizer(object):
  """tf.train.MomentumOptimizer(learning_rate, momentum, use_nesterov=False) [TF-sem, 1.14: training/momentum.py,
  kernels/training_ops.cc ApplyMomentum]:  accum <- accum * momentum + grad ;  var <- var - learning_rate * accum.
  The accumulators are slot variables named '<variable>/Momentum', zero-initialised.  compute_gradients differentiates
  the scalar with respect to tf.trainable_variables() (here: torch autograd over the eager ops of this module, every one
  of which is a differentiable torch op; tf.stop_gradient detaches); apply_gradients also increments global_step."""

  def __init__(self, learning_rate, momentum, use_locking=False, name='Momentum', use_nesterov=False):
    assert not use_nesterov
    self.lr, self.momentum = learning_rate, momentum

  def compute_gradients(self, loss, var_list=None, **_):
    vs = var_list if var_list is not None else trainable_variables()
    for v in vs:
      assert v.t.requires_grad, 'shim_state().track_grad must be set before the variables are created / assigned'
    grads = torch.autograd.grad(_t(loss), [v.t for v in vs], allow_unused=True)
    return [(Tensor(g.detach(), float32) if g is not None else None, v) for g, v in zip(grads, vs)]

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    lr = float(_t(self.lr))
    _G.last_grads = OrderedDict()
    for g, v in grads_and_vars:
      if g is None:
        continue
      gt = _t(g).detach()
      _G.last_grads[v.name] = gt.clone()
      key = v.name + '/Momentum'
      acc = _G.opt_slots.get(key)
      if acc is None:
        acc = torch.zeros_like(gt)
      acc = acc * float(self.momentum) + gt
      _G.opt_slots[key] = acc
      v.assign(v.t.detach() - lr * acc)
    if global_step is not None:
      global_step.assign(global_step.t + 1)
    return None

  def minimize(self, loss, global_step=None, **_):
    return self.apply_gradients(self.compute_gradients(loss), global_step)


train.get_or_create_global_step = _get_or_create_global_step
train.get_global_step = _get_or_create_global_step
train.MomentumOptimizer = _MomentumOptimizer


# ---- tf.estimator: just the names resnet_model_fn touches -----------------------------------------------------
class _ModeKeys(object):
  TRAIN, EVAL, PREDICT = 'train', 'eval', 'infer'


class _EstimatorSpec(object):
  def __init__(self, mode=None, predictions=None, loss=None, train_op=None, eval_metric_ops=None, export_outputs=None, **kw):
    self.mode, self.predictions, self.loss, self.train_op = mode, predictions, loss, train_op
    self.eval_metric_ops, self.export_outputs = eval_metric_ops, export_outputs


estimator = _Namespace(ModeKeys=_ModeKeys, EstimatorSpec=_EstimatorSpec,
                       export=_Namespace(PredictOutput=lambda outputs=None: outputs))


# ---- tf.metrics (streaming) -----------------------------------------------------------------------------------
def _metric_variable(shape, dtype, validate_shape=True, name=None):
  """metrics_impl.metric_variable: a zero-initialised LOCAL variable.  Eager stand-in of "one graph, many session.run
  calls": asking again for the same full name returns the SAME accumulator (the generator re-opens the scope)."""
  full = '/'.join(_G.scope + [name])
  v = _G.metric_vars.get(full)
  if v is None:
    v = Variable(torch.zeros(_ints(shape) if not isinstance(shape, (int, np.integer)) else [int(shape)], dtype=COMPUTE_DTYPE),
                 full, False)
    _G.metric_vars[full] = v
  return v


def _assign_add(ref, value, use_locking=None, name=None):
  ref.t = (ref.t + _t(value).to(ref.t.dtype)).detach()
  return ref


class _Metrics(object):
  """[TF-sem] tf.metrics.mean: total += sum(values), count += size(values), value = total / count (0 when count is 0);
  tf.metrics.accuracy = mean of float(labels == predictions).  Returns (value, update_op): both read AFTER the update
  here (eager)."""
  @staticmethod
  def mean(values, weights=None, metrics_collections=None, updates_collections=None, name=None):
    assert weights is None
    with variable_scope(name, default_name='mean'):
      total = _metric_variable([], float32, name='total')
      count = _metric_variable([], float32, name='count')
    v = _t(values).to(COMPUTE_DTYPE)
    _assign_add(total, v.sum())
    _assign_add(count, torch.tensor(float(v.numel()), dtype=COMPUTE_DTYPE))
    val = Tensor(total.t / count.t if float(count.t) > 0 else torch.zeros((), dtype=COMPUTE_DTYPE), float32)
    return val, val

  @staticmethod
  def accuracy(labels, predictions, weights=None, metrics_collections=None, updates_collections=None, name=None):
    assert weights is None
    p, l = _t(predictions), _t(labels)
    if p.dim() == l.dim() + 1:
      p = p.squeeze(-1)
    if l.dim() == p.dim() + 1:
      l = l.squeeze(-1)
    with variable_scope(name, default_name='accuracy'):
      return _Metrics.mean(Tensor((p.long() == l.long()).to(COMPUTE_DTYPE), float32), name='mean_inner')


metrics = _Metrics()


# ---- tf.image: the tensor part of preprocessing/imagenet_preprocessing.py -----------------------------------------
class _ResizeMethod(object):
  BILINEAR, NEAREST_NEIGHBOR, BICUBIC, AREA = 0, 1, 2, 3


def _resize_bilinear_legacy(t_hwc, out_h, out_w, align_corners):
  """tf.image.resize_images(method=BILINEAR) of TF 1.x = the legacy ResizeBilinear kernel [TF-sem, 1.14:
  core/kernels/resize_bilinear_op.cc + image_resizer_state.h]:
    scale = (align_corners and out > 1) ? (in - 1) / (out - 1) : in / out         (float32)
    in_coord = out_index * scale          (NO half-pixel offset: that only arrived with half_pixel_centers / TF 2)
    lower = floor(in_coord); upper = min(lower + 1, in - 1); lerp = in_coord - lower
    top = tl + (tr - tl) * x_lerp; bottom = bl + (br - bl) * x_lerp; out = top + (bottom - top) * y_lerp
  computed in float32 whatever the input type, no antialiasing when shrinking."""
  dt = COMPUTE_DTYPE
  img = t_hwc.to(dt)
  H, W = int(img.shape[0]), int(img.shape[1])

  def axis(in_size, out_size):
    if align_corners and out_size > 1:
      scale = torch.tensor(float(in_size - 1), dtype=dt) / torch.tensor(float(out_size - 1), dtype=dt)
    else:
      scale = torch.tensor(float(in_size), dtype=dt) / torch.tensor(float(out_size), dtype=dt)
    coord = torch.arange(out_size, dtype=dt) * scale
    lower = torch.floor(coord).long()
    upper = torch.clamp(lower + 1, max=in_size - 1)
    return lower, upper, coord - lower.to(dt)

  ly, uy, fy = axis(H, int(out_h))
  lx, ux, fx = axis(W, int(out_w))
  fy, fx = fy.view(-1, 1, 1), fx.view(1, -1, 1)
  tl, tr = img[ly][:, lx], img[ly][:, ux]
  bl, br = img[uy][:, lx], img[uy][:, ux]
  top = tl + (tr - tl) * fx
  bottom = bl + (br - bl) * fx
  return top + (bottom - top) * fy


class _Image(object):
  ResizeMethod = _ResizeMethod

  @staticmethod
  def resize_images(images, size, method=0, align_corners=False, preserve_aspect_ratio=False):
    assert method == _ResizeMethod.BILINEAR and not preserve_aspect_ratio
    t = _t(images)
    oh, ow = _ints(size) if not isinstance(size, (list, tuple)) else [int(_t(v).item()) if isinstance(v, Tensor) else int(v) for v in size]
    if t.dim() == 3:
      return Tensor(_resize_bilinear_legacy(t, oh, ow, align_corners), float32)
    return Tensor(torch.stack([_resize_bilinear_legacy(e, oh, ow, align_corners) for e in t], 0), float32)

  # The decoded uint8 image stands in for its JPEG bytes: decoding is outside the hot path.
  @staticmethod
  def decode_jpeg(contents, channels=0, dct_method='', **_):
    t = _t(contents)
    assert t.dtype == torch.uint8 and t.dim() == 3
    return Tensor(t, uint8)

  @staticmethod
  def extract_jpeg_shape(contents, **_):
    return Tensor(torch.tensor(list(_t(contents).shape), dtype=torch.int64), int32)

  @staticmethod
  def sample_distorted_bounding_box(image_size, bounding_boxes, min_object_covered=0.1, aspect_ratio_range=None,
                                    area_range=None, max_attempts=None, use_image_if_no_bounding_boxes=None, **_):
    """The random crop box: TF's stream cannot be reproduced, so the shim draws from shim_state().window_rng through the
    generator-supplied sampler `shim_state().box_sampler(height, width, min_object_covered) -> (y, x, h, w)` and
    RECORDS the draw; only the constraints the reference passes (:66-76) are checked here."""
    H, W = _ints(image_size)[:2]
    assert list(aspect_ratio_range) == [0.75, 1.33] and list(area_range) == [0.05, 1.0] and max_attempts == 100
    assert use_image_if_no_bounding_boxes
    y, x, h, w = _G.box_sampler(H, W, float(min_object_covered))
    assert 0 <= y and 0 <= x and h > 0 and w > 0 and y + h <= H and x + w <= W
    _G.window_draws.append(('box', int(y), int(x), int(h), int(w)))
    begin = Tensor(torch.tensor([y, x, 0], dtype=torch.int64), int32)
    size = Tensor(torch.tensor([h, w, -1], dtype=torch.int64), int32)
    return begin, size, None

  @staticmethod
  def decode_and_crop_jpeg(contents, crop_window, channels=0, dct_method='', **_):
    y, x, h, w = _ints(crop_window)
    return Tensor(_t(contents)[y:y + h, x:x + w], uint8)

  @staticmethod
  def random_flip_left_right(image, seed=None):
    flip = builtins_bool(_G.rng.random() < 0.5)     # [TF-sem] uniform draw < 0.5 flips
    _G.window_draws.append(('flip', int(flip)))
    t = _t(image)
    return Tensor(torch.flip(t, [1]) if flip else t, getattr(image, 'dtype', None))

  @staticmethod
  def flip_left_right(image):
    t = _t(image)
    return Tensor(torch.flip(t, [t.dim() - 2]), getattr(image, 'dtype', None))


uint8 = DType('uint8', False, torch.uint8)
image = _Image()


def __getattr__(name):
  """anything the hot path does not touch (tf.estimator, tf.summary, tf.app, ...) is an inert stand-in so that the
  reference's other modules still IMPORT (functions/model_fns.py pulls in the whole run loop)"""
  if name.startswith('__'):
    raise AttributeError(name)
  return mock.MagicMock(name='tf.' + name)
