"""CPU oracle: a restatement of clovaai/assembled-cnn's Assemble-ResNet training path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

**Parity status: wiring PINNED to the reference's source, op arithmetic pinned by stated TF rules.**
The reference ships no test, golden vector or fixture for any hot-path file and
TensorFlow 1.14 cannot be installed here, so no OUTPUT OF TENSORFLOW pins this restatement.  What does pin it:
(a) the reference's own ``nets/resnet_model.py``, ``nets/blocks.py``, ``nets/model_helper.py``,
``functions/model_fns.py``, ``losses/cls_losses.py`` and ``utils/data_util.mixup`` are executed UNMODIFIED under a
torch-backed ``tensorflow`` stand-in (``oracle/tf_shim``; generator ``tests/golden/make_reference_taps.py``, fixture
``tests/golden/reference_taps.json``) and ``tests/test_reference_taps.py`` requires this oracle to reproduce every
variable name / creation order / shape, every named tap and the logits of 8 configurations (all BASELINE ones) to
1e-9 (inference) / 1e-6 (training mode at batch 2) in float64, plus DropBlock, the losses, mixup and the schedules;
(b) the arithmetic INSIDE each tf op (SAME padding, fused-BN moving-variance Bessel correction, average-pool
divisors, label smoothing ...) is restated twice, independently -- here and in the shim -- from the stated TF 1.14
rules marked [TF-sem], with hand-computed known answers in ``tests/test_oracle_semantics.py``; TensorFlow itself
never arbitrates; (c) topology pins (parameter / trainable-tensor counts, ``tests/test_oracle_topology.py``).

Everything here is plain PyTorch-CPU *primitive* ops (conv2d / pad / mean ...)
in fp32 (or fp64) with TF padding / pooling / BN semantics written out
explicitly.  Each function cites the reference ``file:line`` it follows
(paths relative to the reference repo root).

Tensors are NCHW internally; the public ``Model.__call__`` takes NHWC like the
reference pipeline emits (``nets/resnet_model.py:323-327``).  Conv kernels are
stored in TF's HWIO layout ``[k, k, Cin, Cout]``.

``emulate_bf16=True`` rounds activations / weights to bfloat16 at the points
where a bf16 implementation stores them (straight-through gradient), so a
bf16-in / fp32-accumulate implementation can be compared at tight tolerance.
With ``emulate_bf16=False`` this is the literal fp32 graph.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# preprocessing/imagenet_preprocessing.py:46-49
CHANNEL_MEANS = (123.68, 116.78, 103.94)

# functions/data_config.py:44-47
IMAGENET_NUM_CLASSES = 1001
IMAGENET_NUM_TRAIN_IMAGES = 1281167


# --------------------------------------------------------------------------------------
# bf16 storage emulation
# --------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return x.to(torch.bfloat16).to(x.dtype)

  @staticmethod
  def backward(ctx, g):
    return g


def round_bf16(x: torch.Tensor) -> torch.Tensor:
  return _RoundBF16.apply(x)


# --------------------------------------------------------------------------------------
# Variable store (mirrors tf.get_variable creation order under 'resnet_model')
# --------------------------------------------------------------------------------------
def _trunc_normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
  """Normal(0, std) re-sampled into (-2 std, 2 std) like tf.truncated_normal [TF-sem]."""
  out = rng.standard_normal(size=shape)
  bad = np.abs(out) > 2.0
  while bad.any():
    out[bad] = rng.standard_normal(size=int(bad.sum()))
    bad = np.abs(out) > 2.0
  return out * std


class VarStore(object):
  """Ordered variables.  ``trainable`` order == ``tf.trainable_variables()`` order.

  Initialisers (distributional parity only; TF's RNG stream is not reproducible):
  * conv / sk_fc / se / embedding kernels: ``tf.variance_scaling_initializer()`` defaults
    = scale 1.0, fan_in, truncated normal with stddev sqrt(1/fan_in)/0.87962566103423978
    (nets/model_helper.py:77, nets/blocks.py:138,145,173,179) [TF-sem]
  * dense kernel: glorot uniform, bias zeros or -log(C-1) (nets/resnet_model.py:239-247,595-597) [TF-sem]
  * BN: gamma 1 (0 when zero_gamma), beta 0, moving_mean 0, moving_variance 1
    (nets/model_helper.py:30-37) [TF-sem]
  """

  def __init__(self, seed: int = 0, dtype=torch.float32):
    self.rng = np.random.default_rng(seed)
    self.dtype = dtype
    self.trainable: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    self.state: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    self.pending_updates: Dict[str, torch.Tensor] = {}
    self._scope: List[str] = []
    self._counters: Dict[Tuple[str, str], int] = {}

  # ---- scope / naming -------------------------------------------------------------
  def begin_call(self):
    self._scope = []
    self._counters = {}
    self.pending_updates = {}

  def _unique(self, base: str) -> str:
    key = ('/'.join(self._scope), base)
    n = self._counters.get(key, 0)
    self._counters[key] = n + 1
    return base if n == 0 else '%s_%d' % (base, n)

  def push_scope(self, default_name: str):
    """tf.variable_scope(None, default_name): uniquified within the parent scope."""
    self._scope.append(self._unique(default_name))

  def pop_scope(self):
    self._scope.pop()

  def _full(self, layer: str, var: str) -> str:
    return '/'.join(['resnet_model'] + self._scope + [layer, var])

  # ---- variable getters -----------------------------------------------------------
  def _get(self, table, name, init_fn, requires_grad):
    self.last_name = name
    if name not in table:
      t = torch.as_tensor(np.asarray(init_fn(), dtype=np.float64)).to(self.dtype)
      if requires_grad:
        t.requires_grad_(True)
      table[name] = t
    return table[name]

  def conv_kernel(self, k: int, cin: int, cout: int, layer_name: Optional[str] = None) -> torch.Tensor:
    layer = self._unique('conv2d') if layer_name is None else layer_name
    fan_in = k * k * cin
    std = math.sqrt(1.0 / fan_in) / .87962566103423978
    return self._get(self.trainable, self._full(layer, 'kernel'),
                     lambda: _trunc_normal(self.rng, (k, k, cin, cout), std), True)

  def bn_vars(self, c: int, zero_gamma: bool, layer_name: Optional[str] = None):
    layer = self._unique('batch_normalization') if layer_name is None else layer_name
    gamma = self._get(self.trainable, self._full(layer, 'gamma'),
                      lambda: np.zeros(c) if zero_gamma else np.ones(c), True)
    beta = self._get(self.trainable, self._full(layer, 'beta'), lambda: np.zeros(c), True)
    mm_name = self._full(layer, 'moving_mean')
    mv_name = self._full(layer, 'moving_variance')
    mm = self._get(self.state, mm_name, lambda: np.zeros(c), False)
    mv = self._get(self.state, mv_name, lambda: np.ones(c), False)
    return gamma, beta, mm, mv, mm_name, mv_name

  def dense_vars(self, cin: int, cout: int, bias_init: float):
    layer = self._unique('dense')
    limit = math.sqrt(6.0 / (cin + cout))
    kernel = self._get(self.trainable, self._full(layer, 'kernel'),
                       lambda: self.rng.uniform(-limit, limit, size=(cin, cout)), True)
    bias = self._get(self.trainable, self._full(layer, 'bias'),
                     lambda: np.full((cout,), bias_init), True)
    return kernel, bias

  # ---- BN moving-stat updates (tf.GraphKeys.UPDATE_OPS, nets/optimizer_setting.py:36-37)
  def apply_updates(self):
    for name, val in self.pending_updates.items():
      self.state[name] = val.detach().clone()
    self.pending_updates = {}

  def num_params(self) -> int:
    return sum(int(v.numel()) for v in self.trainable.values())


# --------------------------------------------------------------------------------------
# nets/model_helper.py
# --------------------------------------------------------------------------------------
class Ctx(object):
  """Per-call context: variable store + bf16 emulation hooks + named taps."""

  def __init__(self, store: VarStore, emulate_bf16: bool = False):
    self.vs = store
    self.emulate_bf16 = emulate_bf16
    self.taps: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    # optional per-layer record: kernel name -> conv input; gamma name -> (fused BN group output, residual or None)
    self.rec_conv_in: Optional[Dict[str, torch.Tensor]] = None
    self.rec_bn: Optional[Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]]] = None
    # rec_live: keep the LIVE autograd tensors (with retain_grad) instead of detached copies, so that after
    # loss.backward() every recorded activation carries d loss / d activation in .grad (teacher-forced backward checks);
    # rec_extra: further named activations (the output of an SK unit, keyed by the gamma of its 3x3 convolution's BN)
    self.rec_live = False
    self.rec_extra: Optional[Dict[str, torch.Tensor]] = None

  def _keep(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
      return None
    if not self.rec_live:
      return t.detach()
    if t.requires_grad and not t.is_leaf:
      t.retain_grad()
    return t

  def note_conv(self, x: torch.Tensor):
    if self.rec_conv_in is not None:
      self.rec_conv_in[self.vs.last_name] = self._keep(x)

  def note_bn(self, gamma_name: str, out: torch.Tensor, residual: Optional[torch.Tensor]):
    if self.rec_bn is not None:
      self.rec_bn[gamma_name] = (self._keep(out), self._keep(residual))

  def note_extra(self, key: str, t: torch.Tensor):
    if self.rec_extra is not None:
      self.rec_extra[key] = self._keep(t)

  def q(self, x: torch.Tensor) -> torch.Tensor:
    """Storage rounding of an activation."""
    return round_bf16(x) if self.emulate_bf16 else x

  def qw(self, w: torch.Tensor) -> torch.Tensor:
    """fp32 master -> low-precision cast at use (nets/resnet_model.py:286-290)."""
    return round_bf16(w) if self.emulate_bf16 else w

  def tap(self, name: str, x_nchw: torch.Tensor):
    self.taps[name] = x_nchw


def fixed_padding(inputs: torch.Tensor, kernel_size: int) -> torch.Tensor:
  """nets/model_helper.py:40-64 -- zero pad (k-1)//2 before, the rest after (NCHW)."""
  pad_total = kernel_size - 1
  pad_beg = pad_total // 2
  pad_end = pad_total - pad_beg
  return F.pad(inputs, (pad_beg, pad_end, pad_beg, pad_end))


def _conv_raw(x: torch.Tensor, w_hwio: torch.Tensor, kernel_size: int, strides: int) -> torch.Tensor:
  """nets/model_helper.py:67-78.  stride>1: explicit fixed_padding + VALID;
  stride 1: SAME, which for stride 1 is pad_total = k-1 split (k-1)//2 before [TF-sem]."""
  x = fixed_padding(x, kernel_size)  # identical split for both branches
  w = w_hwio.permute(3, 2, 0, 1)  # HWIO -> OIHW
  return F.conv2d(x, w, stride=strides)


def conv2d_fixed_padding(ctx: Ctx, inputs, filters, kernel_size, strides, layer_name=None):
  cin = inputs.shape[1]
  w = ctx.vs.conv_kernel(kernel_size, cin, filters, layer_name)
  name = ctx.vs.last_name
  ctx.note_conv(inputs)
  y = ctx.q(_conv_raw(inputs, ctx.qw(w), kernel_size, strides))
  ctx.note_extra('conv_out:' + name, y)
  return y


def _bn_raw(ctx: Ctx, inputs, training, zero_gamma, momentum, epsilon, layer_name=None):
  """nets/model_helper.py:26-37 (tf.layers.batch_normalization, fused=True).

  [TF-sem] train: normalise with the biased batch variance; the moving variance is fed the
  Bessel-corrected one; ``momentum`` weights the OLD moving value.  Returns fp32 math result
  (no storage rounding) so callers can fuse add / relu before rounding.
  """
  c = inputs.shape[1]
  gamma, beta, mm, mv, mm_name, mv_name = ctx.vs.bn_vars(c, zero_gamma, layer_name)
  ctx.last_gamma_name = mm_name[:-len('moving_mean')] + 'gamma'
  red = [d for d in range(inputs.dim()) if d != 1]
  shape = [1, c] + [1] * (inputs.dim() - 2)
  if training:
    mean = inputs.mean(dim=red)
    var = ((inputs - mean.view(shape)) ** 2).mean(dim=red)
    n = inputs.numel() // c
    unbiased = var * (float(n) / max(n - 1, 1))
    ctx.vs.pending_updates[mm_name] = (mm * momentum + mean.detach() * (1.0 - momentum))
    ctx.vs.pending_updates[mv_name] = (mv * momentum + unbiased.detach() * (1.0 - momentum))
  else:
    mean, var = mm, mv
  inv = torch.rsqrt(var + epsilon)
  return (inputs - mean.view(shape)) * (inv * gamma).view(shape) + beta.view(shape)


def batch_norm(ctx: Ctx, inputs, training, zero_gamma=False, momentum=0.997, epsilon=1e-5,
               relu=False, residual=None, layer_name=None):
  """batch_norm (+ optional residual add, + optional ReLU) with ONE storage rounding.

  The reference applies these as separate ops (nets/resnet_model.py:50-55,92-95); in fp32 the
  result is identical.  Under bf16 emulation a single rounding after the fused group mirrors an
  implementation that never materialises the intermediates.
  """
  y = _bn_raw(ctx, inputs, training, zero_gamma, momentum, epsilon, layer_name)
  if residual is not None:
    y = y + residual
  if relu:
    y = F.relu(y)
  y = ctx.q(y)
  ctx.note_bn(ctx.last_gamma_name, y, residual)
  return y


# --------------------------------------------------------------------------------------
# TF pooling semantics
# --------------------------------------------------------------------------------------
def _same_pad(in_size: int, k: int, s: int) -> Tuple[int, int, int]:
  """[TF-sem] SAME: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2."""
  out = -(-in_size // s)
  total = max((out - 1) * s + k - in_size, 0)
  return out, total // 2, total - total // 2


def max_pool_same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
  """tf.layers.max_pooling2d(k, s, 'SAME') (nets/resnet_model.py:421-424): 112 -> pad 0 before, 1 after."""
  _, ph0, ph1 = _same_pad(x.shape[2], k, s)
  _, pw0, pw1 = _same_pad(x.shape[3], k, s)
  x = F.pad(x, (pw0, pw1, ph0, ph1), value=float('-inf'))
  return F.max_pool2d(x, k, s)


def avg_pool_valid(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
  """average_pooling2d(k, s, 'VALID') on an already zero-padded tensor: divisor always k*k."""
  return F.avg_pool2d(x, k, s)


def avg_pool_same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
  """[TF-sem] SAME average pool divides by the number of VALID (non-pad) elements."""
  _, ph0, ph1 = _same_pad(x.shape[2], k, s)
  _, pw0, pw1 = _same_pad(x.shape[3], k, s)
  xp = F.pad(x, (pw0, pw1, ph0, ph1))
  ones = F.pad(torch.ones_like(x[:1, :1]), (pw0, pw1, ph0, ph1))
  num = F.avg_pool2d(xp, k, s) * (k * k)
  den = F.avg_pool2d(ones, k, s) * (k * k)
  return num / den


def upsample2x_nearest(x: torch.Tensor) -> torch.Tensor:
  """tf.keras.layers.UpSampling2D((2,2)) (nets/resnet_model.py:499): nearest-neighbour repeat."""
  return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


# --------------------------------------------------------------------------------------
# nets/blocks.py
# --------------------------------------------------------------------------------------
_BINOMIAL = {1: [1.], 2: [1., 1.], 3: [1., 2., 1.], 4: [1., 3., 3., 1.], 5: [1., 4., 6., 4., 1.],
             6: [1., 5., 10., 10., 5., 1.], 7: [1., 6., 15., 20., 15., 6., 1.]}


def blur_filter(filt_size: int, dtype=torch.float32) -> torch.Tensor:
  """nets/blocks.py:58-76 -- outer(a, a) built in the activation dtype, then / its sum."""
  a = torch.tensor(_BINOMIAL[filt_size], dtype=torch.float64)
  f = (a[:, None] * a[None, :]).to(dtype)
  return f / f.sum()


def anti_aliased_downsample(ctx: Ctx, inp: torch.Tensor, filt_size=3, stride=2) -> torch.Tensor:
  """nets/blocks.py:45-107: REFLECT pad int((k-1)/2), depthwise binomial filter, stride, VALID."""
  pad = int(1. * (filt_size - 1) / 2)
  c = inp.shape[1]
  if filt_size == 1:
    # nets/blocks.py:79-84 (pad_off == 0): plain subsample
    return inp[:, :, ::stride, ::stride]
  filt = blur_filter(filt_size, inp.dtype).to(inp.dtype)
  x = F.pad(inp, (pad, pad, pad, pad), mode='reflect')
  w = filt.view(1, 1, filt_size, filt_size).repeat(c, 1, 1, 1)
  return ctx.q(F.conv2d(x, w, stride=stride, groups=c))


def sk_conv2d(ctx: Ctx, inputs, filters, strides, training, r=2, L=32, bn_momentum=0.997):
  """nets/blocks.py:110-154."""
  vs = ctx.vs
  x = conv2d_fixed_padding(ctx, inputs, filters * 2, 3, strides)
  x = batch_norm(ctx, x, training, momentum=bn_momentum, relu=True)
  sk_key = ctx.last_gamma_name
  f0, f1 = x[:, :filters], x[:, filters:]                   # tf.split(axis=channel) :130
  fea_u = f0 + f1                                            # :131
  fea_s = ctx.q(fea_u.mean(dim=(2, 3), keepdim=True))        # :134
  d = max(int(filters / r), L)                               # :136
  vs.push_scope('sk_block')
  w1 = vs.conv_kernel(1, filters, d, layer_name='sk_fc_1')
  ctx.note_conv(fea_s)
  fea_z = ctx.q(_conv_raw(fea_s, ctx.qw(w1), 1, 1))
  ctx.note_extra('conv_out:' + vs.last_name, fea_z)
  fea_z = batch_norm(ctx, fea_z, training, momentum=bn_momentum, relu=True)
  w2 = vs.conv_kernel(1, d, filters * 2, layer_name='sk_fc_2')
  att = _conv_raw(fea_z, ctx.qw(w2), 1, 1)                    # logits kept fp32
  vs.pop_scope()
  a = torch.softmax(torch.stack([att[:, :filters], att[:, filters:]], dim=0), dim=0)  # :150-151
  fea_v = ctx.q(f0 * a[0] + f1 * a[1])                       # :152
  ctx.note_extra('sk_out:' + sk_key, fea_v)
  return fea_v


def se_block(ctx: Ctx, x, ratio=16):
  """nets/blocks.py:156-184."""
  vs = ctx.vs
  c = x.shape[1]
  vs.push_scope('se_block')
  squeeze = ctx.q(x.mean(dim=(2, 3), keepdim=True))
  w1 = vs.conv_kernel(1, c, c // ratio, layer_name='seblock_dense_1')
  e = ctx.q(F.relu(_conv_raw(squeeze, ctx.qw(w1), 1, 1)))
  w2 = vs.conv_kernel(1, c // ratio, c, layer_name='seblock_dense_2')
  e = torch.sigmoid(_conv_raw(e, ctx.qw(w2), 1, 1))
  vs.pop_scope()
  return x * e  # caller rounds after the residual add


def generalized_mean_pooling(x, p=3):
  """nets/blocks.py:22-42 (GeM)."""
  n = float(x.shape[2] * x.shape[3])
  eps = 1e-6
  x = torch.clamp(x, eps, 1e12)
  s = torch.clamp((x ** p).sum(dim=(2, 3), keepdim=True), min=eps)
  return (n ** (-1.0 / p)) * s ** (1.0 / p)


def dropblock(x, keep_prob, block_size, gamma_scale=1.0, is_training=True, uniform=None):
  """nets/blocks.py:191-251.  ``uniform`` supplies tf.random_uniform's draw of shape
  [1, C, H-bs+1, W-bs+1] (the mask is shared across the batch, :224,:228)."""
  if not is_training:
    return x
  if (isinstance(keep_prob, float) and keep_prob == 1) or gamma_scale == 0:
    return x
  br = (block_size - 1) // 2
  tl = (block_size - 1) - br
  _, c, h, w = x.shape
  gamma = (1. - keep_prob) * (w * h) / (block_size ** 2) / ((w - block_size + 1) * (h - block_size + 1))
  gamma = gamma_scale * gamma
  if uniform is None:
    raise ValueError('dropblock oracle needs the uniform draw passed in')
  mask = F.relu(torch.sign(gamma - uniform))                # _bernoulli :187-188
  mask = F.pad(mask, (tl, br, tl, br))
  _, p0, p1 = _same_pad(h, block_size, 1)
  mask = F.max_pool2d(F.pad(mask, (p0, p1, p0, p1), value=float('-inf')), block_size, 1)
  mask = 1 - mask
  ret = x * mask
  norm = mask.numel() / (mask.sum() + 1e-8)
  return ret * norm


# --------------------------------------------------------------------------------------
# nets/resnet_model.py
# --------------------------------------------------------------------------------------
def _bottleneck_block_v1(ctx: Ctx, inputs, filters, training, projection_shortcut, strides,
                         zero_gamma=False, dropblock_fn=None, se_block_fn=None, use_sk_block=False,
                         bn_momentum=0.997, anti_alias_filter_size=0, anti_alias_type="",
                         last_relu=True, block_expansion=4):
  """nets/resnet_model.py:35-97."""
  shortcut = inputs
  if projection_shortcut is not None:
    shortcut = projection_shortcut(inputs)
    shortcut = batch_norm(ctx, shortcut, training, momentum=bn_momentum)
    if dropblock_fn:
      shortcut = dropblock_fn(shortcut)

  x = conv2d_fixed_padding(ctx, inputs, filters, 1, 1)
  if dropblock_fn:
    x = batch_norm(ctx, x, training, momentum=bn_momentum)
    x = F.relu(dropblock_fn(x))
  else:
    x = batch_norm(ctx, x, training, momentum=bn_momentum, relu=True)

  s3 = 1 if 'sconv' in anti_alias_type else strides
  if use_sk_block:
    x = sk_conv2d(ctx, x, filters, s3, training, bn_momentum=bn_momentum)
    if dropblock_fn:
      x = dropblock_fn(x)
  else:
    x = conv2d_fixed_padding(ctx, x, filters, 3, s3)
    if dropblock_fn:
      x = batch_norm(ctx, x, training, momentum=bn_momentum)
      x = F.relu(dropblock_fn(x))
    else:
      x = batch_norm(ctx, x, training, momentum=bn_momentum, relu=True)

  if 'sconv' in anti_alias_type and strides != 1:
    x = anti_aliased_downsample(ctx, x, filt_size=anti_alias_filter_size, stride=strides)

  x = conv2d_fixed_padding(ctx, x, block_expansion * filters, 1, 1)
  if dropblock_fn or se_block_fn:
    x = batch_norm(ctx, x, training, zero_gamma=zero_gamma, momentum=bn_momentum)
    bn3_key = ctx.last_gamma_name
    if dropblock_fn:
      x = dropblock_fn(x)
    if se_block_fn:
      x = se_block_fn(x)
    x = x + shortcut
    if last_relu:
      x = F.relu(x)
    x = ctx.q(x)
    ctx.note_extra('block_out:' + bn3_key, x)
    return x
  return batch_norm(ctx, x, training, zero_gamma=zero_gamma, momentum=bn_momentum,
                    residual=shortcut, relu=last_relu)


def block_layer(ctx: Ctx, inputs, filters, bottleneck, block_fn, num_blocks, strides, training, name,
                zero_gamma=False, use_resnet_d=False, dropblock_fn=None, se_block_fn=None,
                use_sk_block=False, bn_momentum=0.997, anti_alias_filter_size=0, anti_alias_type="",
                expansion=4, use_bl=False, last_relu=True):
  """nets/resnet_model.py:99-163."""
  filters_out = filters * expansion if bottleneck else filters

  def projection_shortcut(x):  # :107-121
    if 'proj' in anti_alias_type and strides != 1:
      x = anti_aliased_downsample(ctx, x, filt_size=anti_alias_filter_size, stride=strides)
      return conv2d_fixed_padding(ctx, x, filters_out, 1, 1)
    return conv2d_fixed_padding(ctx, x, filters_out, 1, strides)

  def resnet_d_projection_shortcut(x):  # :123-131
    if strides > 1:
      x = fixed_padding(x, 2)
      x = ctx.q(avg_pool_valid(x, 2, strides))
    else:
      x = ctx.q(avg_pool_same(x, 2, strides))
    return conv2d_fixed_padding(ctx, x, filters_out, 1, 1)

  def bl_projection_shortcut(x):  # :133-141
    if strides > 1:
      x = fixed_padding(x, 3)
      x = ctx.q(avg_pool_valid(x, 3, strides))
    return conv2d_fixed_padding(ctx, x, filters_out, 1, 1)

  if use_resnet_d:
    projection_shortcut_fn = resnet_d_projection_shortcut
  elif use_bl:
    projection_shortcut_fn = bl_projection_shortcut
  else:
    projection_shortcut_fn = projection_shortcut

  # :151-155 -- note: last_relu is NOT forwarded to the first block (defaults to True)
  x = block_fn(ctx, inputs, filters, training, projection_shortcut_fn, strides,
               zero_gamma=zero_gamma, dropblock_fn=dropblock_fn, se_block_fn=se_block_fn,
               use_sk_block=use_sk_block, bn_momentum=bn_momentum,
               anti_alias_filter_size=anti_alias_filter_size, anti_alias_type=anti_alias_type,
               block_expansion=expansion)
  for i in range(1, num_blocks):  # :157-161 -- no anti-alias args for the identity blocks
    x = block_fn(ctx, x, filters, training, None, 1, zero_gamma,
                 dropblock_fn=dropblock_fn, se_block_fn=se_block_fn, use_sk_block=use_sk_block,
                 bn_momentum=bn_momentum, block_expansion=expansion,
                 last_relu=last_relu if i == num_blocks - 1 else True)
  ctx.tap(name, x)
  return x


def get_block_sizes(resnet_size, resnet_version=1):
  """functions/model_fns.py:98-135."""
  if resnet_version == 2:
    choices = {50: [3, 4, 6, 3], 101: [4, 8, 18, 3], 152: [5, 12, 30, 3]}
  else:
    choices = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 200: [3, 24, 36, 3]}
  try:
    return choices[resnet_size]
  except KeyError:
    raise ValueError('Could not find layers for selected Resnet size.\n'
                     'Size received: {}; sizes allowed: {}.'.format(resnet_size, choices.keys()))


class Model(object):
  """functions/model_fns.py:138-198 (ImageNet defaults) over nets/resnet_model.py:166-599."""

  def __init__(self, resnet_size, data_format=None, num_classes=None, resnet_version=1,
               dtype='fp32', no_downsample=False, zero_gamma=False, use_se_block=False,
               use_sk_block=False, bn_momentum=0.997, embedding_size=0, anti_alias_filter_size=0,
               anti_alias_type="", pool_type='gap', loss_type='softmax', bl_alpha=2, bl_beta=4,
               seed=0, emulate_bf16=False, param_dtype=torch.float32):
    if resnet_version not in (1, 2):  # nets/resnet_model.py:200-203
      raise ValueError('Resnet version should be 1 or 2. See README for citations.')
    if resnet_size < 50:  # functions/model_fns.py:160-163, nets/resnet_model.py:205-212
      raise NotImplementedError
    if dtype not in ('fp32', 'fp16', 'bf16'):  # nets/resnet_model.py:214-215
      raise ValueError('dtype must be one of fp32/fp16/bf16')
    self.resnet_size = resnet_size
    self.resnet_version = resnet_version
    self.num_classes = num_classes
    self.num_filters = 64
    self.kernel_size = 7
    self.conv_stride = 2
    self.first_pool_size = 3
    self.first_pool_stride = 2
    self.block_sizes = get_block_sizes(resnet_size, resnet_version)
    self.block_strides = [2, 2, 1, 2] if resnet_version == 2 else [1, 2, 2, 2]
    if no_downsample:
      self.block_strides[-1] = 1
    self.zero_gamma = zero_gamma
    self.use_se_block = use_se_block
    self.use_sk_block = use_sk_block
    self.bn_momentum = bn_momentum
    self.embedding_size = embedding_size
    self.anti_alias_filter_size = anti_alias_filter_size
    self.anti_alias_type = anti_alias_type
    self.pool_type = pool_type
    if pool_type not in ('gap', 'gem', 'flatten'):
      raise NotImplementedError
    self.alpha = bl_alpha
    self.beta = bl_beta
    if loss_type == 'softmax':  # nets/resnet_model.py:238-249
      self.dense_bias_init = 0.0
    elif loss_type in ('sigmoid', 'focal', 'anchor'):
      self.dense_bias_init = -math.log(num_classes - 1)
    else:
      raise NotImplementedError
    self.emulate_bf16 = emulate_bf16
    self.vars = VarStore(seed, param_dtype)
    self.taps: "OrderedDict[str, torch.Tensor]" = OrderedDict()

  # -------------------------------------------------------------------------------------
  def __call__(self, inputs_nhwc, training, reuse=False, use_resnet_d=False, keep_prob=1.0,
               return_embedding=False, dropblock_uniforms=None, record_layers=False, record_live=False):
    """nets/resnet_model.py:305-599.  ``inputs_nhwc``: [N, H, W, 3].  ``record_layers`` keeps every conv input and
    every fused BN-group output (NCHW) in ``self.layer_record`` for the teacher-forced per-layer product checks;
    with ``record_live`` they are the live autograd tensors (retain_grad), so after a backward pass each carries
    d loss / d activation in ``.grad``, and ``self.extra_record`` holds the SK unit outputs."""
    vs = self.vars
    vs.begin_call()
    ctx = Ctx(vs, self.emulate_bf16)
    if record_layers:
      ctx.rec_conv_in, ctx.rec_bn, ctx.rec_extra = OrderedDict(), OrderedDict(), OrderedDict()
      ctx.rec_live = bool(record_live)
    self.layer_record = (ctx.rec_conv_in, ctx.rec_bn)
    self.extra_record = ctx.rec_extra
    bnm = self.bn_momentum
    x = ctx.q(inputs_nhwc.permute(0, 3, 1, 2))
    nf = self.num_filters

    if use_resnet_d and self.resnet_version == 1:  # :328-341
      x = conv2d_fixed_padding(ctx, x, nf // 2, 3, self.conv_stride)
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
      x = conv2d_fixed_padding(ctx, x, nf // 2, 3, 1)
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
      x = conv2d_fixed_padding(ctx, x, nf, 3, 1)
    elif use_resnet_d and self.resnet_version == 2:  # :342-358
      vs.push_scope('stage0')
      x = conv2d_fixed_padding(ctx, x, nf // 2, 3, self.conv_stride)
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
      x = conv2d_fixed_padding(ctx, x, nf // 2, 3, 1)
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
      x = conv2d_fixed_padding(ctx, x, nf, 3, 1)
      vs.pop_scope()
    elif self.resnet_version == 2:  # :359-363
      vs.push_scope('stage0')
      x = conv2d_fixed_padding(ctx, x, nf, self.kernel_size, self.conv_stride)
      vs.pop_scope()
    else:  # :364-367
      x = conv2d_fixed_padding(ctx, x, nf, self.kernel_size, self.conv_stride)
    ctx.tap('initial_conv', x)

    if self.resnet_version == 1:  # :375-377
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
    else:  # :378-381
      vs.push_scope('stage0')
      x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
      vs.pop_scope()

    if self.first_pool_size:
      if self.resnet_version == 2:  # blModule0 :384-419
        vs.push_scope('stage0/pool')
        big0 = conv2d_fixed_padding(ctx, x, nf, 3, 2)
        big0 = batch_norm(ctx, big0, training, momentum=bnm)
        l0 = conv2d_fixed_padding(ctx, x, nf // self.alpha, 3, 1)
        l0 = batch_norm(ctx, l0, training, momentum=bnm, relu=True)
        l0 = conv2d_fixed_padding(ctx, l0, nf // self.alpha, 3, 2)
        l0 = batch_norm(ctx, l0, training, momentum=bnm, relu=True)
        l0 = conv2d_fixed_padding(ctx, l0, nf, 1, 1)
        # relu(big0 + BN(little0)) :413 -- fused into the BN (one storage rounding)
        x = batch_norm(ctx, l0, training, momentum=bnm, residual=big0, relu=True)
        x = conv2d_fixed_padding(ctx, x, nf, 1, 1)
        x = batch_norm(ctx, x, training, momentum=bnm, relu=True)
        vs.pop_scope()
      else:  # :420-425
        x = max_pool_same(x, self.first_pool_size, self.first_pool_stride)
        ctx.tap('initial_max_pool', x)

    se_fn = (lambda t: se_block(ctx, t, ratio=16)) if self.use_se_block else None
    # DropBlock draws: an iterable of [1, C, H-6, W-6] tensors, or a callable shape -> tensor
    db_iter = (iter(dropblock_uniforms) if (dropblock_uniforms is not None and not callable(dropblock_uniforms))
               else None)

    def make_dropblock(gamma_scale):
      if not training or (isinstance(keep_prob, float) and keep_prob == 1.0):
        return None  # blocks.dropblock returns x unchanged (:208-213)

      def fn(t):
        if callable(dropblock_uniforms):
          u = dropblock_uniforms((1, t.shape[1], t.shape[2] - 6, t.shape[3] - 6))
        else:
          u = next(db_iter)
        # one storage rounding where a low-precision implementation stores the DropBlock output (the reference's fp16 graph
        # holds it as an fp16 tensor between ops); the ReLU that may follow commutes with the rounding
        return ctx.q(dropblock(t, keep_prob, 7, gamma_scale, True, u))
      return fn

    for i, num_blocks in enumerate(self.block_sizes):
      num_filters = nf * (2 ** i)
      if i == 2:
        dropblock_fn = make_dropblock(0.25)
      elif i == 3:
        dropblock_fn = make_dropblock(1.0)
      else:
        dropblock_fn = None
      common = dict(zero_gamma=self.zero_gamma, dropblock_fn=dropblock_fn, se_block_fn=se_fn,
                    use_sk_block=self.use_sk_block, bn_momentum=bnm,
                    anti_alias_filter_size=self.anti_alias_filter_size,
                    anti_alias_type=self.anti_alias_type)

      if self.resnet_version == 2 and i < 3:  # :455-516
        vs.push_scope('stage{}'.format(i + 1))
        vs.push_scope('big{}'.format(i + 1))
        big = block_layer(ctx, x, num_filters, True, _bottleneck_block_v1, num_blocks - 1, 2,
                          training, 'big{}'.format(i + 1), last_relu=False, use_bl=True, **common)
        vs.pop_scope()
        vs.push_scope('little{}'.format(i + 1))
        little = block_layer(ctx, x, num_filters // self.alpha, True, _bottleneck_block_v1,
                             max(1, num_blocks // self.beta - 1), 1, training,
                             'little{}'.format(i + 1), use_bl=True, **common)
        little_e = conv2d_fixed_padding(ctx, little, num_filters * 4, 1, 1)
        little_e = _bn_raw(ctx, little_e, training, False, bnm, 1e-5)  # :495-496 (unrounded)
        merge_gamma = ctx.last_gamma_name
        vs.pop_scope()
        big_e = upsample2x_nearest(big)                                # :499
        x = ctx.q(F.relu(little_e + big_e))                            # :501, one storage rounding
        ctx.note_bn(merge_gamma, x, big)                               # residual recorded at its own (half) resolution
        vs.push_scope('merge{}'.format(i + 1))
        x = block_layer(ctx, x, num_filters, True, _bottleneck_block_v1, 1, self.block_strides[i],
                        training, 'merge{}'.format(i + 1), use_bl=True, **common)
        vs.pop_scope()
        vs.pop_scope()
      elif self.resnet_version == 2 and i == 3:  # :518-534
        vs.push_scope('stage{}'.format(i + 1))
        x = block_layer(ctx, x, num_filters, True, _bottleneck_block_v1, num_blocks,
                        self.block_strides[i], training, 'block_layer{}'.format(i + 1),
                        use_resnet_d=use_resnet_d, use_bl=True, **common)
        vs.pop_scope()
      else:  # :536-549
        x = block_layer(ctx, x, num_filters, True, _bottleneck_block_v1, num_blocks,
                        self.block_strides[i], training, 'block_layer{}'.format(i + 1),
                        use_resnet_d=use_resnet_d, **common)

    # head :555-599
    if self.pool_type == 'gap':
      x = ctx.q(x.mean(dim=(2, 3), keepdim=True))
    elif self.pool_type == 'gem':
      x = ctx.q(generalized_mean_pooling(x))
    else:  # flatten: NHWC flatten order (tf.layers.flatten on channels_last)
      x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, 1, 1)
    ctx.tap('final_reduce_mean', x)

    if self.embedding_size > 0:  # :574-586
      w = vs.conv_kernel(1, x.shape[1], self.embedding_size, layer_name='embedding_dense')
      ctx.note_conv(x)
      e = ctx.q(_conv_raw(x, ctx.qw(w), 1, 1))
      # :591-592 applies tf.nn.relu to the squeezed embedding unless it is returned; relu commutes with the storage
      # rounding, so it is folded into the batch-norm group here (one recorded group output, as the product computes it)
      e = batch_norm(ctx, e, training, momentum=bnm, layer_name='embedding_dense_batch_normalization',
                     relu=not return_embedding)
      squeezed = e.flatten(1)
    else:
      squeezed = x.flatten(1)
    if return_embedding:
      self.taps = ctx.taps
      return squeezed
    kernel, bias = vs.dense_vars(squeezed.shape[1], self.num_classes, self.dense_bias_init)
    logits = squeezed @ ctx.qw(kernel) + bias  # fp32 logits (nets/run_loop_classification.py:123)
    ctx.tap('final_dense', logits)
    self.taps = ctx.taps
    return logits

  # convenience -------------------------------------------------------------------------
  def trainable_variables(self) -> "OrderedDict[str, torch.Tensor]":
    return self.vars.trainable

  def taps_nhwc(self) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in self.taps.items():
      out[k] = v.permute(0, 2, 3, 1) if v.dim() == 4 else v
    return out


# --------------------------------------------------------------------------------------
# preprocessing / mixup / losses / optimizer / LR schedule
# --------------------------------------------------------------------------------------
def mean_image_subtraction(image_nhwc: torch.Tensor) -> torch.Tensor:
  """preprocessing/imagenet_preprocessing.py:122-155 (no std division)."""
  means = torch.tensor(CHANNEL_MEANS, dtype=image_nhwc.dtype)
  return image_nhwc - means


def mixup(x, y, lam1, keep_batch_size=True, y_t=None, lam2=None):
  """utils/data_util.py:97-158 with the Beta(alpha, alpha) draws passed in (``lam1`` [, ``lam2``]).

  Preserves the reference quirk at :154: with keep_batch_size the second-half teacher mix uses
  ``y1`` (hard labels), not ``y1_t``.
  """
  b2 = x.shape[0] // 2
  x1, x2 = x[:b2], x[b2:]
  y1, y2 = y[:b2], y[b2:]
  lx = lam1.view(b2, 1, 1, 1)
  ly = lam1.view(b2, 1)
  mixed_x = lx * x1 + (1. - lx) * x2
  mixed_y = ly * y1 + (1. - ly) * y2
  mixed_t = None
  if y_t is not None:
    y1_t, y2_t = y_t[:b2], y_t[b2:]
    mixed_t = ly * y1_t + (1. - ly) * y2_t
  if keep_batch_size:
    lx2 = lam2.view(b2, 1, 1, 1)
    ly2 = lam2.view(b2, 1)
    x3 = torch.flip(x2, [0])
    y3 = torch.flip(y2, [0])
    mixed_x = torch.cat([mixed_x, lx2 * x1 + (1. - lx2) * x3], 0)
    mixed_y = torch.cat([mixed_y, ly2 * y1 + (1. - ly2) * y3], 0)
    if y_t is not None:
      y3_t = torch.flip(y2_t, [0])
      mixed_t = torch.cat([mixed_t, ly2 * y1 + (1. - ly2) * y3_t], 0)  # sic (:154)
  return mixed_x.detach(), mixed_y.detach(), (mixed_t.detach() if mixed_t is not None else None)


def softmax_cross_entropy(logits, onehot_labels, label_smoothing=0.0):
  """tf.losses.softmax_cross_entropy(weights=1.0) [TF-sem]: smooth y(1-e)+e/C; mean over batch."""
  c = onehot_labels.shape[1]
  if label_smoothing > 0:
    onehot_labels = onehot_labels * (1.0 - label_smoothing) + label_smoothing / c
  logp = torch.log_softmax(logits, dim=1)
  return -(onehot_labels * logp).sum(dim=1).mean()


def get_sup_loss(logits, onehot_labels, cls_loss_type='softmax', label_smoothing=0.0):
  """losses/cls_losses.py:23-41."""
  if cls_loss_type == 'softmax':
    return softmax_cross_entropy(logits, onehot_labels, label_smoothing)
  if cls_loss_type == 'sigmoid':
    ce = F.binary_cross_entropy_with_logits(logits, onehot_labels, reduction='none')
    return ce.sum() / onehot_labels.sum()
  raise AssertionError('cross_entropy is None')


def kd_loss(logits, teacher_labels, kd_temp):
  """nets/run_loop_classification.py:156-162: T^2 * CE(logits / T, softmax(teacher / T))."""
  return kd_temp * kd_temp * softmax_cross_entropy(logits / kd_temp, teacher_labels)


def l2_loss(trainable: "OrderedDict[str, torch.Tensor]", weight_decay: float):
  """nets/run_loop_classification.py:166-178: wd * sum(l2_loss(v)) over names w/o 'batch_normalization'."""
  tot = 0.0
  for name, v in trainable.items():
    if 'batch_normalization' not in name:
      tot = tot + 0.5 * (v.float() ** 2).sum()
  return weight_decay * tot


def total_loss(model: Model, logits, onehot_labels, *, label_smoothing=0.0, cls_loss_type='softmax',
               kd_temp=0.0, teacher_labels=None, weight_decay=0.0):
  """CE + L2 + KD (nets/run_loop_classification.py:144-179).  Returns (loss, parts dict)."""
  ce = get_sup_loss(logits.float(), onehot_labels, cls_loss_type, label_smoothing)
  kd = kd_loss(logits.float(), teacher_labels, kd_temp) if kd_temp > 0 else torch.zeros(())
  l2 = l2_loss(model.trainable_variables(), weight_decay)
  return ce + l2 + kd, {'cross_entropy': ce, 'l2_loss': l2, 'cross_entropy_kd': kd}


def split_kd_labels(labels, kd_temp):
  """nets/run_loop_classification.py:90-96."""
  half = labels.shape[1] // 2
  onehot, teacher_logits = labels[:, :half], labels[:, half:]
  return onehot, torch.softmax(teacher_logits / kd_temp, dim=1)


def momentum_step(params: List[torch.Tensor], grads: List[torch.Tensor], accums: List[torch.Tensor],
                  lr: float, momentum: float):
  """tf.train.MomentumOptimizer, non-Nesterov [TF-sem]: a <- m*a + g ; w <- w - lr*a
  (nets/optimizer_setting.py:29-34)."""
  with torch.no_grad():
    for p, g, a in zip(params, grads, accums):
      a.mul_(momentum).add_(g)
      p.sub_(lr * a)


def learning_rate_with_decay(learning_rate_decay_type, batch_size, batch_denom, num_images,
                             num_epochs_per_decay, learning_rate_decay_factor, end_learning_rate,
                             piecewise_lr_boundary_epochs, piecewise_lr_decay_rates, base_lr,
                             warmup_epochs=0, train_epochs=None) -> Callable[[int], float]:
  """functions/model_fns.py:36-95 (host scalar)."""
  initial = base_lr * batch_size / batch_denom
  batches_per_epoch = num_images / batch_size
  decay_steps = int(batches_per_epoch * num_epochs_per_decay)

  def fn(global_step: int) -> float:
    warmup_steps = int(batches_per_epoch * warmup_epochs)
    if warmup_steps > 0 and global_step < warmup_steps:
      return initial * float(global_step) / float(warmup_steps)
    gs = global_step - warmup_steps
    t = learning_rate_decay_type
    if t == 'exponential':  # staircase
      return initial * learning_rate_decay_factor ** math.floor(gs / decay_steps)
    if t == 'fixed':
      return base_lr
    if t == 'polynomial':
      s = min(gs, decay_steps)
      return (initial - end_learning_rate) * (1 - s / decay_steps) + end_learning_rate
    if t == 'piecewise':
      bounds = [int(batches_per_epoch * e) for e in piecewise_lr_boundary_epochs]
      vals = [initial * float(d) for d in piecewise_lr_decay_rates]
      for b, v in zip(bounds, vals):
        if global_step <= b:
          return v
      return vals[-1]
    if t == 'cosine':
      total = int(batches_per_epoch * train_epochs) - warmup_steps
      s = min(gs, total)
      return 0.5 * (1 + math.cos(math.pi * s / total)) * initial
    raise NotImplementedError
  return fn


def keep_prob_decay(starter_kp, end_kp, decay_steps) -> Callable[[int], float]:
  """functions/model_fns.py:26-33 (polynomial_decay power 1, no cycle)."""
  def fn(global_step):
    s = min(global_step, decay_steps)
    return (starter_kp - end_kp) * (1 - s / decay_steps) + end_kp
  return fn


# --------------------------------------------------------------------------------------
# One training step (nets/run_loop_classification.py:60-234 + nets/optimizer_setting.py)
# --------------------------------------------------------------------------------------
class TrainState(object):
  def __init__(self, model: Model):
    self.model = model
    self.accums: Optional[List[torch.Tensor]] = None
    self.global_step = 0


def train_step(state: TrainState, images_nhwc, labels, *, lr, momentum=0.9, weight_decay=1e-4,
               label_smoothing=0.0, kd_temp=0.0, mixup_type=0, lam1=None, lam2=None,
               use_resnet_d=False, loss_scale=1.0, num_classes=IMAGENET_NUM_CLASSES,
               cls_loss_type='softmax', grad_hook=None):
  """One optimisation step; returns dict(loss, parts, logits, grads)."""
  m = state.model
  if kd_temp > 0:
    onehot, teacher = split_kd_labels(labels, kd_temp)
  else:
    onehot = F.one_hot(labels.long(), num_classes).to(images_nhwc.dtype)
    teacher = None
  x = images_nhwc
  if mixup_type == 1:
    x, onehot, teacher = mixup(x, onehot, lam1, keep_batch_size=False, y_t=teacher)
  elif mixup_type == 2:
    x, onehot, teacher = mixup(x, onehot, lam1, keep_batch_size=True, y_t=teacher, lam2=lam2)
  logits = m(x, True, use_resnet_d=use_resnet_d)
  loss, parts = total_loss(m, logits, onehot, label_smoothing=label_smoothing,
                           cls_loss_type=cls_loss_type, kd_temp=kd_temp, teacher_labels=teacher,
                           weight_decay=weight_decay)
  params = list(m.trainable_variables().values())
  grads = torch.autograd.grad(loss * loss_scale, params, allow_unused=True)
  grads = [(g / loss_scale if g is not None else torch.zeros_like(p)) for g, p in zip(grads, params)]
  if grad_hook is not None:
    grads = grad_hook(grads)
  if state.accums is None:
    state.accums = [torch.zeros_like(p) for p in params]
  momentum_step(params, grads, state.accums, lr, momentum)
  m.vars.apply_updates()
  state.global_step += 1
  return {'loss': loss.detach(), 'parts': {k: v.detach() if torch.is_tensor(v) else v for k, v in parts.items()},
          'logits': logits.detach(), 'grads': grads, 'mixed_images': x, 'onehot': onehot, 'teacher': teacher}
