"""Host-side runtime of the training path: flat parameter arenas, a hand-written backward tape and
the layer objects (conv+BN groups, SK / SE units, pools) the model walker composes.

There is no autograd: every layer records an explicit backward closure that launches the matching
HIP kernels (conv dgrad / wgrad, BN backward, ...), because the fused kernels (conv + BN statistics,
BN-apply + residual + ReLU) do not map 1:1 onto autograd nodes.

Layouts: activations NHWC bf16; conv / fc / dense kernels KRSC ([Cout][R][S][Cin]) with an fp32
master copy, a bf16 shadow (fprop operand) and a bf16 CRSK copy (dgrad operand); BN gamma/beta and
all gradients fp32.  All trainable tensors live in ONE flat fp32 arena (plus same-shaped gradient /
momentum / bf16 arenas) so the optimiser and the gradient all-reduce are single flat launches.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

BN_EPS = 1e-5  # nets/model_helper.py:26
# ASM_LAZY_DZ=0: materialise the masked shortcut gradient dz in the BN backward apply (the pre-round-2 path; A/B runs)
LAZY_DZ = os.environ.get('ASM_LAZY_DZ', '1') != '0'
# ASM_BN_DEFER=0: a projection shortcut's batch norm is applied by its own pass (materialised) instead of inside the add
DEFER_BN = os.environ.get('ASM_BN_DEFER', '1') != '0'


def dual_bn_on() -> bool:
  """ASM_BN_DUAL=0: two separate batch-norm backwards for a projection block (A/B runs, tests); cached until ops.refresh_tuning()"""
  return ops.knob('ASM_BN_DUAL', '1') != '0'


def _round_up(n: int, m: int) -> int:
  return (n + m - 1) // m * m


# ---------------------------------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------------------------------
class ParamSpec(object):
  __slots__ = ('name', 'shape', 'numel', 'decay', 'init', 'offset', 'index')

  def __init__(self, name, shape, decay, init, index):
    self.name = name
    self.shape = tuple(int(s) for s in shape)
    self.numel = int(np.prod(self.shape))
    self.decay = decay
    self.init = init
    self.offset = -1
    self.index = index


class ParamArena(object):
  """Trainable variables in creation order (== tf.trainable_variables() order of the reference) mapped
  onto flat device arenas.  The weight-decayed set (every variable whose name lacks
  'batch_normalization', nets/run_loop_classification.py:166-177) is stored first so weight decay is a
  per-segment scalar of the optimiser launch."""

  def __init__(self):
    self.specs: "OrderedDict[str, ParamSpec]" = OrderedDict()
    self.state_specs: "OrderedDict[str, Tuple[Tuple[int, ...], float, int]]" = OrderedDict()
    self.finalized = False
    self.decay_elems = 0
    self.total_elems = 0
    self.w32 = self.g32 = self.m32 = self.w16 = self.state = None
    self.derived: List[Callable[[], None]] = []  # refresh hooks (CRSK copies, stem packing)
    self.on_grad: Optional[Callable[[int], None]] = None  # dp.GradSync.notify (gradient-ready watermark)
    self._held: List[int] = []                            # notifications queued between hold_grads() and pass_grads()
    self._holding = False
    self._deferred: List[int] = []
    self._deferring = False
    self.wt_specs: List[dict] = []   # CRSK (dgrad operand) copies: one flat bf16 arena, one batched launch
    self.wt16 = None
    self._wt_table = None
    # Weight gradients are leaves of the backward graph (only the optimiser / all-reduce consume them), so they run
    # on a second HIP stream beside the dgrad -> BN-backward chain: MFMA-bound wgrad blocks and HBM-bound
    # normalisation kernels share the CUs.  None = everything on the compute stream.
    self.side_stream = None
    self._sides = []
    self.compute_stream = None       # the stream the running backward pass was started on (set by Model.backward)
    self.extra_streams = []          # other streams gradient work may run on (the model's BigLittle branch stream)

  def notify_grad(self, name: str):
    """The gradient slot of ``name`` has been enqueued (on the compute stream, or on the weight-gradient stream
    for conv kernels); backward order is the reverse of creation order, so every slot above it in its segment has
    been enqueued too."""
    if self.on_grad is not None:
      if self._deferring:
        self._deferred.append(self.specs[name].offset)
      elif self._holding:
        self._held.append(self.specs[name].offset)
      else:
        self.on_grad(self.specs[name].offset)

  # A projection block runs its shortcut branch's backward BEFORE its main branch's (the pooled gradient then rides in
  # conv1's input-gradient epilogue), but the shortcut's variables were created first, so the gradient-ready watermark
  # must not see them before the main branch's: hold_grads() queues notifications (shortcut branch), pass_grads() lets
  # them through again (main branch), release_grads() replays the queue (after the block).
  def hold_grads(self):
    self._holding = self.on_grad is not None

  def pass_grads(self):
    self._holding = False

  def release_grads(self):
    held, self._held, self._holding = self._held, [], False
    if self.on_grad is not None:
      for off in held:
        self.on_grad(off)

  # The big branch of a BigLittle stage runs its backward on a second stream BESIDE the little branch's (model.py): its
  # variables were created before the little branch's, so while the two are interleaved its notifications are deferred
  # (a list of their own: the projection blocks inside either branch keep using hold / pass / release) and replayed, in
  # the order they were made, once the little branch is through and the streams are joined.
  def defer_grads(self, on: bool):
    self._deferring = bool(on) and self.on_grad is not None

  def flush_deferred(self):
    deferred, self._deferred, self._deferring = self._deferred, [], False
    if self.on_grad is not None:
      for off in deferred:
        self.on_grad(off)

  def enable_side_stream(self, fresh: bool = False):
    """``fresh``: drop parked streams and create new ones instead of reusing them."""
    if fresh:
      self._parked = []
    if self.w32 is not None and self.w32.is_cuda and self.side_stream is None and getattr(self, '_parked', None):
      self._sides, self._parked = self._parked, []       # the very streams that were switched off
      self._side_rr = 0
      self.side_stream = self._sides[0]
      return
    if self.w32 is not None and self.w32.is_cuda and self.side_stream is None:
      # ASM_WGRAD_STREAMS=n: n side streams taken round robin, so that consecutive weight gradients (leaves of the
      # backward graph, independent of each other) may also overlap each other.  Same box, ms per step: 1 stream 27.09 /
      # 27.13, 2 streams 26.96 / 26.96, 3 streams 27.07; no side stream 27.40.
      n = max(1, int(ops.knob('ASM_WGRAD_STREAMS', '2')))
      self._sides = [torch.cuda.Stream(device=self.w32.device) for _ in range(n)]
      self._side_rr = 0
      self.side_stream = self._sides[0]

  def pick_side_stream(self):
    if self.side_stream is None:
      return None
    s = self._sides[self._side_rr % len(self._sides)]
    self._side_rr += 1
    return s

  def join_side_stream(self):
    """Make the compute stream wait for every weight gradient enqueued so far."""
    if self.side_stream is not None:
      for s in self._sides:
        ops.stream_join(torch.cuda.current_stream(), s)

  def disable_side_stream(self):
    """Weight gradients back onto the compute stream.  The stream objects are parked, not dropped: switching the side
    streams on again reuses them."""
    self.join_side_stream()
    if self._sides:
      self._parked = list(self._sides)
    self.side_stream, self._sides = None, []

  def join_all_streams(self, into=None):
    """Make ``into`` (default: the CURRENT stream) wait for everything enqueued so far on every stream this model launches
    gradient work on (the compute stream of the running backward pass -- or the current stream when no backward pass set
    one --, the weight-gradient streams, the BigLittle branch stream): what dp.GradSync does before it hands a bucket to
    RCCL, whose stream is ordered against the stream that is current at the call only."""
    if self.w32 is None or not self.w32.is_cuda:
      return
    cur = torch.cuda.current_stream()
    dst = cur if into is None else into
    srcs = [self.compute_stream if self.compute_stream is not None else cur] + list(self._sides) + list(self.extra_streams)
    if into is not None and cur not in srcs:
      srcs.append(cur)
    for s in srcs:
      if s is not None and s != dst:
        ops.stream_join(dst, s)

  def register(self, name, shape, decay, init) -> ParamSpec:
    if self.finalized:
      if name not in self.specs:
        raise RuntimeError('parameter %s requested after the model was built' % name)
      return self.specs[name]
    if name not in self.specs:
      self.specs[name] = ParamSpec(name, shape, decay, init, len(self.specs))
    return self.specs[name]

  def register_state(self, name, shape, fill: float):
    if name not in self.state_specs:
      if self.finalized:
        raise RuntimeError('state %s requested after the model was built' % name)
      self.state_specs[name] = (tuple(shape), fill, -1)

  def finalize(self, device, seed: int):
    rng = np.random.default_rng(seed)
    off = 0
    for decay_pass in (True, False):
      for sp in self.specs.values():
        if sp.decay == decay_pass:
          sp.offset = off
          off += _round_up(sp.numel, 8)
      if decay_pass:
        self.decay_elems = off
    self.total_elems = off
    host = np.zeros((off,), dtype=np.float32)
    for sp in self.specs.values():  # creation order => reproducible RNG stream
      host[sp.offset:sp.offset + sp.numel] = np.asarray(sp.init(rng), dtype=np.float32).reshape(-1)
    self.w32 = torch.from_numpy(host).to(device)
    self.g32 = torch.zeros_like(self.w32)
    self.m32 = torch.zeros_like(self.w32)
    self.w16 = torch.empty((off,), dtype=torch.bfloat16, device=device)
    soff = 0
    shost = []
    new_specs = OrderedDict()
    for name, (shape, fill, _) in self.state_specs.items():
      n = int(np.prod(shape))
      new_specs[name] = (shape, fill, soff)
      shost.append(np.full((_round_up(n, 8),), fill, dtype=np.float32))
      soff += _round_up(n, 8)
    self.state_specs = new_specs
    self.state = torch.from_numpy(np.concatenate(shost) if shost else np.zeros((0,), np.float32)).to(device)
    woff = 0
    rows = []
    begin = tiles = 0
    for ws in self.wt_specs:
      ws['offset'] = woff
      sp = self.specs[ws['name']]
      rows.append([sp.offset, woff, ws['K'], ws['RS'], ws['C'], ws['ldk'], begin, tiles])
      begin += sp.numel
      tiles += ws['RS'] * (-(-ws['K'] // 64)) * (-(-ws['C'] // 64))
      woff += _round_up(ws['C'] * ws['RS'] * ws['ldk'], 8)
    self._wt_total = begin
    self._wt_tiles = tiles
    self.wt16 = torch.zeros((max(woff, 8),), dtype=torch.bfloat16, device=device)
    self._wt_table = torch.tensor(rows if rows else [[0] * 8], dtype=torch.int32).to(device)
    self.finalized = True
    self.refresh_shadows()

  # views ------------------------------------------------------------------------------------------
  def _view(self, arena, name):
    sp = self.specs[name]
    return arena[sp.offset:sp.offset + sp.numel].view(sp.shape)

  def w(self, name):
    return self._view(self.w32, name)

  def g(self, name):
    return self._view(self.g32, name)

  def m(self, name):
    return self._view(self.m32, name)

  def wb(self, name):
    return self._view(self.w16, name)

  def st(self, name):
    shape, _, off = self.state_specs[name]
    n = int(np.prod(shape))
    return self.state[off:off + n].view(shape)

  def refresh_shadows(self):
    """fp32 master -> bf16 shadow, then the derived layouts (after init / weight import)."""
    ops.cast_f32_to_bf16(self.w32, self.w16)
    self.refresh_derived()

  def alloc_wt(self, name, K, RS, Cn, ldk) -> dict:
    ws = dict(name=name, K=K, RS=RS, C=Cn, ldk=ldk, offset=-1)
    self.wt_specs.append(ws)
    return ws

  def wt_view(self, ws) -> torch.Tensor:
    n = ws['C'] * ws['RS'] * ws['ldk']
    return self.wt16[ws['offset']:ws['offset'] + n]

  def refresh_derived(self):
    if self.wt_specs:
      ops.filter_transpose_tiled(self.w16, self.wt16, self._wt_table, len(self.wt_specs), self._wt_tiles)
    for fn in self.derived:
      fn()

  def num_params(self) -> int:
    return sum(sp.numel for sp in self.specs.values())


# initialisers (distributional parity with TF) -----------------------------------
def variance_scaling_init(shape_krsc):
  """tf.variance_scaling_initializer() defaults: fan_in, truncated normal, stddev sqrt(1/fan_in)/0.8796."""
  k, r, s, c = shape_krsc
  std = math.sqrt(1.0 / (r * s * c)) / .87962566103423978

  def init(rng):
    out = rng.standard_normal(size=shape_krsc)
    bad = np.abs(out) > 2.0
    while bad.any():
      out[bad] = rng.standard_normal(size=int(bad.sum()))
      bad = np.abs(out) > 2.0
    return out * std
  return init


def glorot_uniform_init(shape_kc):
  k, c = shape_kc
  limit = math.sqrt(6.0 / (k + c))
  return lambda rng: rng.uniform(-limit, limit, size=shape_kc)


def const_init(shape, value):
  return lambda rng: np.full(shape, value, dtype=np.float32)


# ---------------------------------------------------------------------------------------------------
# activations + tape
# ---------------------------------------------------------------------------------------------------
class Var(object):
  """An activation: NHWC bf16 tensor (``data`` is None during the shape-only build walk).

  The gradient may be held LAZILY MASKED: (``_grad``, ``grad_mask``) stands for _grad * [mask bit set] -- the masked
  gradient dz = dy * [y > 0] that the ReLU behind a residual add sends down the shortcut branch.  The two consumers
  that dominate (the input gradient of the next 1x1 convolution, which adds it in its epilogue, and the batch-norm
  backward of a projection shortcut) read (dy, mask) directly, so dz is never written; anything else just reads
  ``.grad``, which materialises it."""
  __slots__ = ('_data', 'shape', '_grad', 'grad_mask', 'grad_owned', 'needs_grad', 'bn_ctx', 'red_ctx', 'red',
               'pre_dy', 'deferred', 'pool_grad')

  def __init__(self, data, shape=None, needs_grad=True):
    self._data = data
    # A ReLU-less conv + batch-norm output whose normalisation has NOT been applied yet: (y, M, C, scale, shift).  The
    # block-final layer that adds it as the shortcut applies both batch norms in one pass (ops.bn_apply_dual); any other
    # reader just touches .data, which runs the ordinary apply pass once.
    self.deferred = None
    self.shape = tuple(data.shape) if data is not None else tuple(shape)
    self._grad = None
    self.grad_mask = None
    self.grad_owned = False
    self.needs_grad = needs_grad
    # projection-shortcut fusion: the output of a ReLU-less conv + batch norm carries what its BN backward needs (bn_ctx);
    # the block-final layer that adds it behind one ReLU then runs BOTH batch-norm backwards in one reduce + one apply
    # (ops.bn_bwd_dual) and leaves this layer's dy here (pre_dy)
    self.bn_ctx = None
    self.pre_dy = None
    # a gradient contribution still in pooled form: (dpool [N,Hp,Wp,C], k, stride, pad, count_valid) stands for
    # avgpool_bwd(dpool); a 1x1 stride-1 convolution reading this activation gathers it in its input-gradient epilogue
    # (asm_conv2d_dgrad_pooled), anything else reads .grad, which scatters it through asm_avgpool_bwd
    self.pool_grad = None
    # batch-norm backward sums in the producing input gradient's epilogue (asm_conv2d_dgrad_bnred).  red_ctx, set in the forward
    # pass on the output of a conv -> BN [-> + shortcut] [-> ReLU] layer: (its pre-BN convolution output y, its packed ReLU mask
    # or None) -- what a convolution that reads this activation needs to reduce (sum dz, sum dz * y) while it writes the
    # activation's gradient.  red, set by that convolution's backward: (partials, the gradient tensor they were reduced from).
    # Anything that changes the gradient afterwards (another fan-in term, a lazy mask) drops it; the layer's own backward uses it
    # only if the gradient it finds IS that tensor, unmasked -- otherwise it runs its reduce pass as before.
    self.red_ctx = None
    self.red = None

  @property
  def data(self):
    if self._data is None and self.deferred is not None:
      y, M, Cn, scale, shift = self.deferred
      self._data = ops.bn_apply(y, M, Cn, scale, shift, None, 0, False)
    return self._data

  @data.setter
  def data(self, t):
    self._data = t

  @property
  def grad(self):
    if self.grad_mask is not None:
      self._grad = ops.mask_apply(self._grad, self.grad_mask)
      self.grad_mask = None
      self.grad_owned = True
      self.red = None
    if self.pool_grad is not None:
      dp, k, stride, pad, cv = self.pool_grad
      self.pool_grad = None
      self.red = None
      if self._grad is None:
        self._grad = ops.avgpool_bwd(dp, self.shape, k, stride, pad, cv)
      elif self.grad_owned:
        ops.avgpool_bwd(dp, self.shape, k, stride, pad, cv, addend=self._grad)      # in place
      else:
        self._grad = ops.add_bf16(self._grad, ops.avgpool_bwd(dp, self.shape, k, stride, pad, cv))
      self.grad_owned = True
    return self._grad

  @grad.setter
  def grad(self, g):
    self._grad = g
    self.grad_mask = None
    self.red = None

  def take_masked_grad(self):
    """-> (gradient tensor, packed mask or None) without materialising the product.  A pooled contribution nobody
    gathered (take_pool_grad) is scattered first, so a caller that bypasses ``.grad`` can never drop it."""
    if self.pool_grad is not None:
      self.grad      # the getter scatters it (and materialises a lazy mask)
    return self._grad, self.grad_mask

  def take_pool_grad(self, d):
    """-> the pending pooled contribution if a convolution with descriptor ``d`` can gather it in its input-gradient
    epilogue (and clear it), else None (a later .grad read scatters it)"""
    if self.pool_grad is None or not ops.dgrad_pool_ok(d):
      return None
    pg, self.pool_grad = self.pool_grad, None
    return pg


def accum_grad(v: Var, g: torch.Tensor, owned: bool, mask: Optional[torch.Tensor] = None):
  """v.grad += g [* mask].  ``owned`` says whether g may later be updated in place by us."""
  if not v.needs_grad:
    return
  v.red = None           # the gradient changes: sums reduced from an earlier form of it are void
  if mask is not None:
    if v._grad is None:
      v._grad, v.grad_mask, v.grad_owned = g, mask, False
      return
    g, owned = ops.mask_apply(g, mask), True
  if v.grad is None:
    v.grad = g
    v.grad_owned = owned
  elif v.grad_owned:
    ops.add_bf16(v.grad, g, out=v.grad)
  else:
    v.grad = ops.add_bf16(v.grad, g)
    v.grad_owned = True


class Ctx(object):
  """Per-call context shared by the walker and the layers."""

  def __init__(self, arena: ParamArena, training: bool, dry: bool, bn_momentum: float, device,
               record_tape: bool, layers: Optional[list] = None):
    self.arena = arena
    self.layers = layers if layers is not None else []
    self._cursor = 0
    self.dlogits = None
    self.uniforms = None      # iterator of DropBlock uniform draws supplied by the caller (tests)
    self.db_static = None     # DropBlockState: draws and gamma in static device buffers (recordable step)
    self.rng = None           # torch.Generator for DropBlock when none are supplied
    self.training = training
    self.dry = dry
    self.bn_momentum = bn_momentum
    self.device = device
    self.tape: Optional[List[Callable[[], None]]] = [] if (record_tape and not dry) else None
    self.taps: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    self._scope: List[str] = []
    self._counters: Dict[Tuple[str, str], int] = {}

  def next_uniform(self, shape):
    """tf.random_uniform draw of one DropBlock call: float32 [H-6, W-6, C]."""
    if self.uniforms is not None:
      u = next(self.uniforms)
      if tuple(u.shape) != tuple(shape):
        raise ValueError('dropblock uniform has shape %s, expected %s' % (tuple(u.shape), tuple(shape)))
      return u.contiguous()
    return torch.rand(shape, generator=self.rng, device=self.device, dtype=torch.float32)

  def layer(self, factory):
    """Layer objects are created once (in walk order, during the shape-only build pass) and re-used."""
    if self.dry:
      obj = factory()
      self.layers.append(obj)
      return obj
    obj = self.layers[self._cursor]
    self._cursor += 1
    return obj

  # TF-style variable naming ------------------------------------------------------------------------
  def unique(self, base: str) -> str:
    key = ('/'.join(self._scope), base)
    n = self._counters.get(key, 0)
    self._counters[key] = n + 1
    return base if n == 0 else '%s_%d' % (base, n)

  def push_scope(self, default_name: str):
    self._scope.append(self.unique(default_name))

  def pop_scope(self):
    self._scope.pop()

  def full_name(self, layer: str, var: str) -> str:
    return '/'.join(['resnet_model'] + self._scope + [layer, var])

  def record(self, fn: Callable[[], None]):
    if self.tape is not None:
      self.tape.append(fn)

  def tap(self, name: str, v: Var):
    if not self.dry:
      self.taps[name] = v.data

  def backward(self):
    for fn in reversed(self.tape):
      fn()
    self.tape = []


# ---------------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------------
class ConvKernel(object):
  """One conv / fc / dense kernel: parameter slot + bf16 shadow + CRSK copy + launches.
  conv2d_fixed_padding (nets/model_helper.py:67-78)."""

  def __init__(self, ctx: Ctx, k: int, cin: int, cout: int, layer_name: Optional[str] = None,
               dense: bool = False, stem: bool = False, need_dgrad: bool = True):
    layer = ctx.unique('dense' if dense else 'conv2d') if layer_name is None else layer_name
    self.name = ctx.full_name(layer, 'kernel')
    self.k, self.cin, self.cout = k, cin, cout
    self.stem = stem           # 3-channel first conv (k in {3,7}, stride 2) over the halo buffer
    self.stem_len = _round_up(4 * k, 8)
    self.need_dgrad = need_dgrad and not stem
    self.arena = ctx.arena
    shape = (cout, k, k, cin)
    init = glorot_uniform_init((cout, cin)) if dense else variance_scaling_init(shape)
    if dense:
      shape_init = init
      init = lambda rng: shape_init(rng).reshape(cout, 1, 1, cin)
    ctx.arena.register(self.name, shape, True, init)
    self.kpad = _round_up(cout, 8)  # dy channel stride for the backward of a non-multiple-of-8 Cout
    self._wpack = None
    self._wts = None
    if not ctx.arena.finalized:
      if self.stem:
        ctx.arena.derived.append(self._refresh)
      if self.need_dgrad:
        self._wts = ctx.arena.alloc_wt(self.name, cout, k * k, cin, self.kpad)

  def _refresh(self):
    a = self.arena
    dev = a.w32.device
    if self.stem:
      if self._wpack is None:
        self._wpack = torch.empty((self.cout, self.k, 1, self.stem_len), dtype=torch.bfloat16, device=dev)
      ops.stem_pack_filter(a.w(self.name), self._wpack, self.cout, self.k)

  # descriptors -------------------------------------------------------------------------------------
  def desc(self, N, H, W, stride, out_f32=False, ldy=0):
    if self.stem:
      # k x k / 2 first conv over the zero-haloed [N][H+6][W+6][4] buffer: R=k, S=1, "C" = stem_len
      # (a row of k pixels x 4 channels is contiguous; 7x7 -> 32 elements, 3x3 -> 16)
      Hp, Wp = H + 6, W + 6
      return ops.make_conv_desc(N, Hp, Wp, self.stem_len, self.cout, self.k, 1, 2, pad=0,
                                Ho=ops.out_size(H, self.k, 2), Wo=ops.out_size(W, self.k, 2),
                                img_pitch=Hp * Wp * 4, row_pitch=Wp * 4, pix_pitch=4, ldy=ldy, out_f32=out_f32)
    return ops.make_conv_desc(N, H, W, self.cin, self.cout, self.k, self.k, stride, ldy=ldy, out_f32=out_f32)

  def weight(self):
    return self._wpack if self.stem else self.arena.wb(self.name)

  def _stem_view(self, x: torch.Tensor) -> torch.Tensor:
    """The halo buffer has 3 zero pixels on every side; a k x k conv needs (k-1)//2, so start the
    view (3 - pad) rows / pixels in (a 16-byte aligned element offset)."""
    off = 3 - (self.k - 1) // 2
    if off == 0:
      return x
    Wp = x.shape[2]
    return x.view(-1)[(off * Wp + off) * 4:]

  def fprop(self, d, x: torch.Tensor, want_stats: bool):
    if self.stem:
      x = self._stem_view(x)
    return ops.conv_fprop(d, x, self.weight(), want_stats)

  def fprop_bn(self, d, x: torch.Tensor, scale, shift, residual, relu):
    """inference: conv + folded BN [+ residual] [+ ReLU] in one launch"""
    if self.stem:
      x = self._stem_view(x)
    return ops.conv_fprop_bn(d, x, self.weight(), scale, shift, residual, relu)

  def _wgrad(self, d, x: torch.Tensor, dy: torch.Tensor):
    a = self.arena
    if self.stem:
      dwp = torch.empty((self.cout, self.k, self.stem_len), dtype=torch.float32, device=x.device)
      ops.conv_wgrad(d, self._stem_view(x), dy, dwp)
      ops.stem_unpack_grad(dwp, a.g(self.name), self.cout, self.k)
    else:
      ops.conv_wgrad(d, x, dy, a.g(self.name))

  def wgrad_streamed(self, d, x: torch.Tensor, dy: torch.Tensor):
    """dW into the gradient arena, on the weight-gradient side stream when there is one"""
    a = self.arena
    side = a.pick_side_stream()
    if side is not None:
      ops.stream_join(side, a.compute_stream or torch.cuda.current_stream())     # x and dy were produced on the compute stream
      if self.stem or ops.timer_on():                    # these allocate / record events through torch: its own context
        with torch.cuda.stream(side):
          self._wgrad(d, x, dy)
      else:
        ops.launch_on(side)
        try:
          self._wgrad(d, x, dy)
        finally:
          ops.launch_on(None)
      x.record_stream(side)                               # keep the caching allocator from recycling them early
      dy.record_stream(side)
    else:
      self._wgrad(d, x, dy)
    a.notify_grad(self.name)     # on the compute stream's side of things: dp.GradSync joins the other streams itself

  def backward(self, d, x: torch.Tensor, dy: torch.Tensor, need_dx: bool,
               addend: Optional[torch.Tensor] = None, addend_mask: Optional[torch.Tensor] = None, pool=None
               ) -> Optional[torch.Tensor]:
    """dW into the gradient arena; returns dx [+ addend [where addend_mask]] (or None)."""
    return self.backward_red(d, x, dy, need_dx, addend, addend_mask, pool, None)[0]

  def backward_red(self, d, x: torch.Tensor, dy: torch.Tensor, need_dx: bool, addend=None, addend_mask=None, pool=None,
                   red_ctx=None):
    """backward() that also reduces the batch-norm backward sums of the layer that produced this convolution's input, in the
    epilogue that writes dx (``red_ctx`` = that layer's (pre-BN output, ReLU mask or None), nn.Var.red_ctx) -> (dx, partials or
    None).  Falls back to the plain input gradient (partials None) wherever the fused form does not apply."""
    a = self.arena
    self.wgrad_streamed(d, x, dy)
    if self.stem:
      return None, None
    if not need_dx:
      return None, None
    if addend_mask is not None and (d.stride != 1 or d.C % 8) and not ops.dgrad_s2_ok(d):
      addend, addend_mask = ops.mask_apply(addend, addend_mask), None     # the strided forms take a plain addend
    dd = d
    if self.kpad != self.cout:  # dy carries kpad channels (zero padded)
      dd = ops.make_conv_desc(d.N, d.H, d.W, d.C, self.kpad, d.R, d.S, d.stride, pad=d.pad, Ho=d.Ho, Wo=d.Wo)
    if red_ctx is not None and pool is None and ops.dgrad_bnred_ok(dd):
      return ops.conv_dgrad_bnred(dd, dy, a.wt_view(self._wts), addend, addend_mask, red_ctx[0], red_ctx[1])
    return ops.conv_dgrad(dd, dy, a.wt_view(self._wts), addend, addend_mask, pool), None


class BatchNorm(object):
  """batch_norm (nets/model_helper.py:26-37): parameters + moving statistics."""

  def __init__(self, ctx: Ctx, c: int, zero_gamma: bool = False, layer_name: Optional[str] = None):
    layer = ctx.unique('batch_normalization') if layer_name is None else layer_name
    self.c = c
    self.arena = ctx.arena
    self.gamma = ctx.full_name(layer, 'gamma')
    self.beta = ctx.full_name(layer, 'beta')
    self.mm = ctx.full_name(layer, 'moving_mean')
    self.mv = ctx.full_name(layer, 'moving_variance')
    ctx.arena.register(self.gamma, (c,), False, const_init((c,), 0.0 if zero_gamma else 1.0))
    ctx.arena.register(self.beta, (c,), False, const_init((c,), 0.0))
    ctx.arena.register_state(self.mm, (c,), 0.0)
    ctx.arena.register_state(self.mv, (c,), 1.0)


def conv_bn(ctx: Ctx, x: Var, conv: ConvKernel, bn: BatchNorm, stride: int, relu: bool,
            residual: Optional[Var] = None, res_mode: int = 0, tap_pre: Optional[str] = None) -> Var:
  """conv2d_fixed_padding -> batch_norm [-> + residual] [-> relu], forward and (taped) backward.

  res_mode 1: residual has the output shape; 2: residual is [N, H/2, W/2, C] and is nearest-upsampled
  (UpSampling2D((2,2)) + add of the BigLittle merge, nets/resnet_model.py:499-501)."""
  if conv.stem:
    N, Hp, Wp, _ = x.shape
    H, W = Hp - 6, Wp - 6
  else:
    N, H, W, _ = x.shape
  d = conv.desc(N, H, W, stride)
  out_shape = (N, d.Ho, d.Wo, conv.cout)
  if ctx.dry:
    return Var(None, out_shape)
  a = ctx.arena
  M = N * d.Ho * d.Wo
  Cn = conv.cout
  gamma, beta = a.w(bn.gamma), a.w(bn.beta)
  taped = ctx.tape is not None
  rm = res_mode if residual is not None else 0
  # the shortcut's batch norm rides in this layer's apply pass when nobody else has asked for its output
  res_def = residual.deferred if (residual is not None and rm == 1 and residual._data is None and ctx.training) else None
  res_t = residual.data if (residual is not None and res_def is None) else None
  # [N, 1, 1, d] squeeze layers (SK / SE fc): the whole BN is one launch per direction instead of 3-4 latency-bound ones
  small = ctx.training and residual is None and d.Ho * d.Wo == 1 and not conv.stem and ops.bn_small_ok(M)
  mask_t = None
  if small:
    y, _ = conv.fprop(d, x.data, False)
    out_t, mask_t, mean, invstd = ops.bn_small_fwd(y, M, Cn, gamma, beta, BN_EPS, ctx.bn_momentum, a.st(bn.mm),
                                                   a.st(bn.mv), relu, want_mask=taped)
  elif ctx.training:
    y, part = conv.fprop(d, x.data, True)
    mean, invstd, scale, shift = ops.bn_finalize(part, M, Cn, gamma, beta, BN_EPS, ctx.bn_momentum,
                                                 a.st(bn.mm), a.st(bn.mv))
  elif tap_pre is None and not taped and rm in (0, 1) and Cn % 8 == 0:
    # inference: moving-statistics BN, the shortcut add and the ReLU ride in the conv epilogue (no BN pass at all)
    scale, shift = ops.bn_infer_coeffs(Cn, gamma, beta, a.st(bn.mm), a.st(bn.mv), BN_EPS)
    return Var(conv.fprop_bn(d, x.data, scale, shift, res_t if rm == 1 else None, relu))
  else:
    y, _ = conv.fprop(d, x.data, False)
    scale, shift = ops.bn_infer_coeffs(Cn, gamma, beta, a.st(bn.mm), a.st(bn.mv), BN_EPS)
    mean = invstd = None
  if tap_pre is not None:
    ctx.taps[tap_pre] = y
  defer = (DEFER_BN and ctx.training and not small and not relu and residual is None and tap_pre is None and Cn % 8 == 0)
  if small:
    pass
  elif res_def is not None and res_def[1] == M and res_def[2] == Cn:
    if taped and relu:
      out_t, mask_t = ops.bn_apply_dual(y, res_def[0], M, Cn, scale, shift, res_def[3], res_def[4], True, want_mask=True)
    else:
      out_t, mask_t = ops.bn_apply_dual(y, res_def[0], M, Cn, scale, shift, res_def[3], res_def[4], relu), None
  elif defer:
    out_t = None
  elif taped and relu:   # keep the 1-bit ReLU mask for the backward kernels (16x less traffic than re-reading out)
    out_t, mask_t = ops.bn_apply(y, M, Cn, scale, shift, residual.data if residual is not None else None, rm, True,
                                 d.Ho, d.Wo, want_mask=True)
  else:
    out_t, mask_t = ops.bn_apply(y, M, Cn, scale, shift, residual.data if residual is not None else None, rm, relu,
                                 d.Ho, d.Wo), None
  out = Var(out_t, out_shape)
  if out_t is None:
    out.deferred = (y, M, Cn, scale, shift)
  if taped and ctx.training and not small and not relu and residual is None:
    out.bn_ctx = (y, gamma, mean, invstd, bn, M, Cn)
  # (not for a block-final layer whose projection shortcut's batch norm shares its backward -- bn_bwd_dual reduces both)
  if (taped and ctx.training and not small and out_t is not None and Cn % 8 == 0 and (mask_t is not None or not relu)
      and not (residual is not None and residual.bn_ctx is not None and dual_bn_on())):
    out.red_ctx = (y, mask_t if relu else None)

  if ctx.tape is not None:
    x_t = x.data

    def bwd():
      if out.pre_dy is not None:      # projection shortcut: the block-final layer already ran this batch norm's backward
        dy, out.pre_dy = out.pre_dy, None
        a.notify_grad(bn.gamma)
        pool = x.take_pool_grad(d) if (conv.kpad == conv.cout and x.needs_grad) else None
        if x.pool_grad is not None:
          x.grad                                        # (not gatherable here) scatter it now
        xg, xmask = x.take_masked_grad()
        dx = conv.backward(d, x_t, dy, x.needs_grad, addend=xg, addend_mask=xmask, pool=pool)
        if dx is not None:
          x.grad, x.grad_owned = dx, True
        return
      if out.pool_grad is not None:
        out.grad                                        # a pooled contribution nobody gathered: scatter it now
      # the incoming gradient may be lazily masked by the ReLU of the block this layer's output was the shortcut of:
      # a BN without its own ReLU takes that mask as if it were its own (dz = dout * mask is exactly what it needs)
      dout, in_mask = out.take_masked_grad()
      if dout is None:
        raise RuntimeError('conv_bn backward: no gradient reached this layer')
      bmask, brelu = (mask_t if relu else None), relu
      if in_mask is not None:
        if relu or small or residual is not None:
          dout, in_mask = out.grad, None                 # (never on this path's networks) materialise
        else:
          bmask, brelu = in_mask, True
      lazy = LAZY_DZ and residual is not None and relu and res_mode == 1 and mask_t is not None and residual.needs_grad
      # BigLittle merge (res_mode 2): the 2x2 block sum reads (dout, mask) as well, so dz is not written there either
      lazy_up = LAZY_DZ and residual is not None and relu and res_mode == 2 and mask_t is not None
      want_dz = residual is not None and relu and not lazy and not lazy_up
      rc = residual.bn_ctx if residual is not None else None
      dual = (lazy and rc is not None and residual._grad is None and residual.pre_dy is None and rc[5] == M and rc[6] == Cn
              and in_mask is None and dual_bn_on())
      if small:
        dy, dz = ops.bn_small_bwd(dout, y, bmask, M, Cn, gamma, mean, invstd, a.g(bn.gamma), a.g(bn.beta)), None
      elif dual:
        # out = relu(bn(y) + bn_sc(y_sc)): both batch norms see the same masked gradient -> one reduce, one apply
        sc_bn = rc[4]
        dy, residual.pre_dy = ops.bn_bwd_dual(dout, y, rc[0], mask_t, M, Cn,
                                              (gamma, mean, invstd, a.g(bn.gamma), a.g(bn.beta)),
                                              (rc[1], rc[2], rc[3], a.g(sc_bn.gamma), a.g(sc_bn.beta)))
        dz = None
      else:
        # the input gradient that wrote dout may have reduced (sum dz, sum dz * y) already (Var.red)
        raw = None
        if (out.red is not None and out.red[1] is dout and in_mask is None and out.red_ctx is not None
            and out.red_ctx[1] is bmask):
          raw = out.red[0]
        out.red = None
        dy, dz = ops.bn_bwd(dout, y, bmask, brelu, M, Cn, gamma, mean, invstd, a.g(bn.gamma), a.g(bn.beta), want_dz,
                            raw_part=raw)
      a.notify_grad(bn.gamma)
      if residual is not None:
        if dual:
          pass             # the shortcut layer's gradient is complete: its dy waits in residual.pre_dy
        elif lazy:           # dz = dout * mask is NOT written: the shortcut branch receives (dout, mask)
          accum_grad(residual, dout, False, mask=mask_t)
        else:
          dres = dz if relu else dout
          if res_mode == 2:
            accum_grad(residual, ops.upsample2x_bwd(dout, mask_t) if lazy_up else ops.upsample2x_bwd(dres), True)
          else:
            accum_grad(residual, dres, relu)
      pool = x.take_pool_grad(d) if (conv.kpad == conv.cout and x.needs_grad and not conv.stem) else None
      if x.pool_grad is not None:
        x.grad                                          # (not gatherable here) scatter it now
      xg, xmask = x.take_masked_grad()
      # fan-in add (and a pending average-pool backward) fused into the dgrad epilogue; and, when x is itself the output of
      # a conv -> BN layer, the reduce pass of THAT batch norm's backward (x.red_ctx)
      dx, part = conv.backward_red(d, x_t, dy, x.needs_grad, addend=xg, addend_mask=xmask, pool=pool, red_ctx=x.red_ctx)
      if dx is not None:
        x.grad, x.grad_owned = dx, True
        if part is not None:
          x.red = (part, dx)
      out.grad = None
    ctx.record(bwd)
  return out


def conv_plain(ctx: Ctx, x: Var, conv: ConvKernel, out_f32: bool, ldy: int = 0):
  """A bare 1x1 conv (SK fc2, SE fc, dense).  Returns (tensor, backward(dy_tensor) -> None)."""
  N, H, W, _ = x.shape
  d = conv.desc(N, H, W, 1, out_f32=out_f32, ldy=ldy)
  if ctx.dry:
    return None, None, (N, d.Ho, d.Wo, ldy if ldy else conv.cout)
  y, _ = conv.fprop(d, x.data, False)
  x_t = x.data

  def bwd(dy: torch.Tensor):
    # dy: bf16 with channel stride == ldy (or cout)
    # the backward descriptor is always a bf16 one; dy's row stride is ldy when the logits were padded
    dd = ops.make_conv_desc(d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride, pad=d.pad, Ho=d.Ho, Wo=d.Wo,
                            ldy=ldy if (ldy and ldy != conv.cout) else 0)
    dx = conv.backward(dd, x_t, dy, x.needs_grad, addend=x.grad)
    if dx is not None:
      x.grad, x.grad_owned = dx, True
  return y, bwd, tuple(y.shape)


class SKUnit(object):
  """blocks.sk_conv2d (nets/blocks.py:110-154)."""

  def __init__(self, ctx: Ctx, cin: int, filters: int, r: int = 2, L: int = 32):
    self.filters = filters
    self.conv = ConvKernel(ctx, 3, cin, filters * 2)
    self.bn = BatchNorm(ctx, filters * 2)
    self.d = max(int(filters / r), L)
    ctx.push_scope('sk_block')
    self.fc1 = ConvKernel(ctx, 1, filters, self.d, layer_name='sk_fc_1')
    self.bn1 = BatchNorm(ctx, self.d)
    self.fc2 = ConvKernel(ctx, 1, self.d, filters * 2, layer_name='sk_fc_2')
    ctx.pop_scope()

  def __call__(self, ctx: Ctx, x: Var, stride: int) -> Var:
    # Training path: the BN + ReLU of the 3x3 convolution is applied on the fly by its three readers (pooled sum,
    # select, their backward twins), so the 2F-channel normalised tensor, its ReLU mask and the gradient df are never
    # written (csrc/sk_fused.hip).  ASM_SK_FUSED=0 keeps the materialising path (A/B runs, inference uses it too:
    # there the conv epilogue already emits f).
    if ctx.training and not ctx.dry and 2 * self.filters <= 2048 and ops.knob('ASM_SK_FUSED', '1') != '0':
      return self._call_fused(ctx, x, stride)
    F_ = self.filters
    f = conv_bn(ctx, x, self.conv, self.bn, stride, relu=True)
    N, H, W, _ = f.shape
    if ctx.dry:
      return Var(None, (N, H, W, F_))
    s = Var(ops.sk_gap(f.data, F_))                                   # mean_hw(f0 + f1)  :131-134
    z = conv_bn(ctx, s, self.fc1, self.bn1, 1, relu=True)            # :137-143
    att, fc2_bwd, _ = conv_plain(ctx, z, self.fc2, out_f32=True)      # :144-148 (fp32 logits)
    v_t = ops.sk_select_fwd(f.data, att, F_)                          # :149-152
    v = Var(v_t)
    if ctx.tape is not None:
      # tape order: [conv_bn(f), conv_bn(z)] were recorded already; backward must run
      #   select_bwd_att -> fc2 bwd -> conv_bn(z) bwd -> select_bwd_f -> conv_bn(f) bwd
      # so the z / f closures are pulled off the tape and re-sequenced here.
      bwd_z = ctx.tape.pop()
      bwd_f = ctx.tape.pop()
      f_t = f.data

      def bwd():
        dv = v.grad
        datt = ops.sk_select_bwd_att(f_t, dv, att, F_)
        fc2_bwd(datt)                      # -> z.grad
        bwd_z()                            # -> s.grad
        df = ops.sk_select_bwd_f(dv, att, s.grad, F_)
        s.grad = None
        accum_grad(f, df, True)
        bwd_f()                            # -> x.grad
        v.grad = None
      ctx.record(bwd)
    return v


  def _call_fused(self, ctx: Ctx, x: Var, stride: int) -> Var:
    F_ = self.filters
    conv, bn, a = self.conv, self.bn, ctx.arena
    N, H, W, _ = x.shape
    d = conv.desc(N, H, W, stride)
    M, C2 = N * d.Ho * d.Wo, 2 * F_
    gamma, beta = a.w(bn.gamma), a.w(bn.beta)
    y, part = conv.fprop(d, x.data, True)                               # conv + fused statistics :115-118
    mean, invstd, scale, shift = ops.bn_finalize(part, M, C2, gamma, beta, BN_EPS, ctx.bn_momentum, a.st(bn.mm),
                                                 a.st(bn.mv))
    # Factorised BN backward (csrc/sk_fused.hip): the pooled-sum pass and the gate-gradient pass also emit per-image
    # statistics, from which the batch-norm reduction follows without another pass over y and dV (1.8 GB of HBM reads per
    # step).  Exact and tested.  Rounds 2-3 kept it off: the two per-image passes run one workgroup per image, and their
    # five cross-lane sums (1024 LDS reads on 8 lanes each) cost more than the removed pass.  With the two-level lane sum
    # of round 4 it wins: 25.73 -> 25.57 ms per step (same box, two rounds each).  ASM_SK_FACTOR=0 turns it off.
    factor = ctx.tape is not None and ops.knob('ASM_SK_FACTOR', '1') == '1'
    if factor:
      s_t, mask_stats = ops.sk_gap_bn(y, scale, shift, F_, mean, invstd)
      s = Var(s_t)
    else:
      s = Var(ops.sk_gap_bn(y, scale, shift, F_))                       # mean_hw(f0 + f1)  :131-134
    z = conv_bn(ctx, s, self.fc1, self.bn1, 1, relu=True)               # :137-143
    att, fc2_bwd, _ = conv_plain(ctx, z, self.fc2, out_f32=True)         # :144-148 (fp32 logits)
    v = Var(ops.sk_select_bn_fwd(y, scale, shift, att, F_))             # :149-152
    if ctx.tape is not None:
      bwd_z = ctx.tape.pop()        # re-sequenced below: select_bwd_att -> fc2 -> (bn1, fc1) -> BN backward -> conv
      x_t = x.data

      def bwd():
        dv = v.grad
        if dv is None:
          raise RuntimeError('sk unit backward: no gradient reached this layer')
        if factor:
          datt, grad_stats = ops.sk_select_bn_bwd_att(y, scale, shift, dv, att, F_, mean, invstd)
        else:
          datt, grad_stats = ops.sk_select_bn_bwd_att(y, scale, shift, dv, att, F_), None
        fc2_bwd(datt)                      # -> z.grad
        bwd_z()                            # -> s.grad
        dy = ops.sk_bn_bwd(dv, att, s.grad, y, scale, shift, gamma, mean, invstd, a.g(bn.gamma), a.g(bn.beta), F_,
                           grad_stats, mask_stats if factor else None)
        s.grad = None
        a.notify_grad(bn.gamma)
        # fan-in add fused into the dgrad epilogue, and the reduce pass of conv1's batch-norm backward (x.red_ctx)
        dx, part = conv.backward_red(d, x_t, dy, x.needs_grad, addend=x.grad, red_ctx=x.red_ctx)
        if dx is not None:
          x.grad, x.grad_owned = dx, True
          if part is not None:
            x.red = (part, dx)
        v.grad = None
      ctx.record(bwd)
    return v


class SEUnit(object):
  """blocks.se_block (nets/blocks.py:156-184)."""

  def __init__(self, ctx: Ctx, c: int, ratio: int = 16):
    self.c = c
    ctx.push_scope('se_block')
    self.fc1 = ConvKernel(ctx, 1, c, c // ratio, layer_name='seblock_dense_1')
    self.fc2 = ConvKernel(ctx, 1, c // ratio, c, layer_name='seblock_dense_2')
    ctx.pop_scope()

  def __call__(self, ctx: Ctx, x: Var) -> Var:
    if ctx.dry:
      return Var(None, x.shape)
    sq = Var(ops.gap_fwd(x.data))
    e1_t, fc1_bwd, _ = conv_plain(ctx, sq, self.fc1, out_f32=False)
    e1r = Var(ops.relu_fwd(e1_t))
    e_t, fc2_bwd, _ = conv_plain(ctx, e1r, self.fc2, out_f32=True)
    y = Var(ops.se_scale_fwd(x.data, e_t))
    if ctx.tape is not None:
      x_t = x.data

      def bwd():
        dy = y.grad
        de = ops.se_scale_bwd_e(x_t, dy, e_t)
        fc2_bwd(de)                                     # -> e1r.grad
        fc1_bwd(ops.relu_bwd(e1r.grad, e1r.data))       # -> sq.grad
        accum_grad(x, ops.se_scale_bwd_x(dy, e_t, sq.grad), True)
        e1r.grad = sq.grad = y.grad = None
      ctx.record(bwd)
    return y


# ---- pools ---------------------------------------------------------------------------------------------
def max_pool_3x3_s2_same(ctx: Ctx, x: Var) -> Var:
  """tf.layers.max_pooling2d(3, 2, 'SAME') (nets/resnet_model.py:421-424)."""
  N, H, W, Cn = x.shape
  if ctx.dry:
    return Var(None, (N, (H + 1) // 2, (W + 1) // 2, Cn))
  y_t, am = ops.maxpool3x3s2_fwd(x.data)
  y = Var(y_t)
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, ops.maxpool3x3s2_bwd(y.grad, am, x.shape), True)
      y.grad = None
    ctx.record(bwd)
  return y


def avg_pool(ctx: Ctx, x: Var, k: int, stride: int, pad: int, count_valid: bool) -> Var:
  """fixed_padding + tf.layers.average_pooling2d of the ResNet-D / BL shortcuts (nets/resnet_model.py:123-141)."""
  N, H, W, Cn = x.shape
  if count_valid:   # SAME, stride 1
    Ho, Wo = H, W
  else:             # zero pad (k-1)//2 before / rest after, then VALID
    Ho, Wo = (H + (k - 1) - k) // stride + 1, (W + (k - 1) - k) // stride + 1
  if ctx.dry:
    return Var(None, (N, Ho, Wo, Cn))
  y = Var(ops.avgpool_fwd(x.data, k, stride, pad, Ho, Wo, count_valid))
  if ctx.tape is not None:
    def bwd():
      if y.grad is None:
        raise RuntimeError('avg_pool backward: no gradient reached this layer')
      if (x.needs_grad and x.pool_grad is None and stride in (1, 2) and stride <= k <= 2 * stride and
          ops.knob('ASM_POOL_FUSE', '1') != '0'):
        # leave the contribution in pooled form: the block's first 1x1 convolution gathers it in its input-gradient
        # epilogue (its backward runs after this one: model._bottleneck orders the tape that way)
        x.pool_grad = (y.grad, k, stride, pad, count_valid)
      elif x.needs_grad and x.grad is not None and x.grad_owned:
        # gradient fan-in of the block input (main path arrived first): add inside the pool backward, in place
        ops.avgpool_bwd(y.grad, x.shape, k, stride, pad, count_valid, addend=x.grad)
      else:
        accum_grad(x, ops.avgpool_bwd(y.grad, x.shape, k, stride, pad, count_valid), True)
      y.grad = None
    ctx.record(bwd)
  return y


def blur_pool(ctx: Ctx, x: Var, k: int, stride: int) -> Var:
  """blocks.anti_aliased_downsample (nets/blocks.py:45-107)."""
  N, H, W, Cn = x.shape
  if k == 1:
    raise NotImplementedError('anti_alias_filter_size=1 hard-codes NCHW slicing in the reference (blocks.py:79-84)')
  if ctx.dry:
    return Var(None, (N, ops.blur_out_size(H, k, stride), ops.blur_out_size(W, k, stride), Cn))
  y = Var(ops.blurpool_fwd(x.data, k, stride))
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, ops.blurpool_bwd(y.grad, x.shape, k, stride), True)
      y.grad = None
    ctx.record(bwd)
  return y


def global_avg_pool(ctx: Ctx, x: Var) -> Var:
  N, H, W, Cn = x.shape
  if ctx.dry:
    return Var(None, (N, 1, 1, Cn))
  y = Var(ops.gap_fwd(x.data))
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, ops.gap_bwd(y.grad, x.shape), True)
      y.grad = None
    ctx.record(bwd)
  return y


def gem_pool(ctx: Ctx, x: Var, p: float = 3.0) -> Var:
  """blocks.generalized_mean_pooling (nets/blocks.py:22-42)."""
  N, H, W, Cn = x.shape
  if ctx.dry:
    return Var(None, (N, 1, 1, Cn))
  y_t, ssum = ops.gem_fwd(x.data, p)
  y = Var(y_t)
  if ctx.tape is not None:
    x_t = x.data

    def bwd():
      accum_grad(x, ops.gem_bwd(x_t, y.grad, ssum, p), True)
      y.grad = None
    ctx.record(bwd)
  return y


def flatten_pool(ctx: Ctx, x: Var) -> Var:
  """pool_type == 'flatten' (nets/resnet_model.py:565-569): NHWC flatten, kept as a [N,1,1,H*W*C] tensor."""
  N, H, W, Cn = x.shape
  if ctx.dry:
    return Var(None, (N, 1, 1, H * W * Cn))
  y = Var(x.data.view(N, 1, 1, H * W * Cn))
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, y.grad.view(N, H, W, Cn), False)
      y.grad = None
    ctx.record(bwd)
  return y


DROPBLOCK_SIZE = 7  # nets/resnet_model.py:434-439


def dropblock_gamma(gamma_scale: float, keep_prob: float, H: int, W: int) -> float:
  """the Bernoulli mean of one DropBlock call (nets/blocks.py:231-232), in the host's double arithmetic"""
  bs = DROPBLOCK_SIZE
  return gamma_scale * (1. - keep_prob) * (W * H) / (bs ** 2) / ((W - bs + 1) * (H - bs + 1))


class DropBlockState(object):
  """DropBlock inputs of a training step in STATIC device buffers, so that the step can be recorded once and replayed
  (train.Trainer.capture) while keep_prob follows its schedule (functions/model_fns.py:26-33,221-226) and the draws change:
  one float32 uniform buffer per DropBlock call of the walk (creation order) and one slot of ``gamma`` per call.  The first
  step discovers the calls (``slot`` allocates and fills as the walk reaches them); from then on ``prepare`` rewrites every
  buffer BEFORE the step -- torch kernels on the step's stream, outside the recorded region -- and the walk only picks
  the buffers up.  The gamma values are the ones the eager path passes by value (same double expression, rounded to
  float32 once), so a recorded step and an eager step with the same draws keep the same mask bit for bit."""
  CAP = 256          # DropBlock calls per step (Assemble-ResNet-152: 4 per block x 39 blocks of stages 3 - 4)
  RING = 16          # pinned staging buffers for the per-step gamma upload

  def __init__(self, device):
    self.device = torch.device(device)
    self.slots = []          # (uniform buffer, gamma_scale, H, W) in creation order
    self.gamma = torch.zeros((self.CAP,), dtype=torch.float32, device=self.device)
    self.known = False       # a whole step has been walked: the call list is final
    self._i = 0
    self._kp = 1.0
    self._pin = None
    self._pin_ev = None
    self._k = 0

  def begin(self, keep_prob: float, uniforms, rng):
    """Before a step: rewrite gamma and the draws (known call list) or arm discovery (first step)."""
    self._i = 0
    self._kp = float(keep_prob)
    self._given = list(uniforms) if uniforms is not None else None
    self._rng = rng
    if not self.known:
      self.slots = []          # a discovery step that raised mid-walk left a partial list: discover from scratch
      return
    n = len(self.slots)
    if self._given is not None and len(self._given) != n:
      raise ValueError('dropblock_uniforms holds %d draws, the step makes %d DropBlock calls' % (len(self._given), n))
    vals = [dropblock_gamma(gs, self._kp, H, W) for (_, gs, H, W) in self.slots]
    if self.gamma.is_cuda:
      if self._pin is None:
        self._pin = [torch.empty((self.CAP,), dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self._pin_ev = [None] * self.RING
      k = self._k
      self._k = (k + 1) % self.RING
      if self._pin_ev[k] is not None:
        self._pin_ev[k].synchronize()      # the upload that last read this staging buffer (RING steps ago) is done
      self._pin[k][:n] = torch.tensor(vals, dtype=torch.float32)
      self.gamma[:n].copy_(self._pin[k][:n], non_blocking=True)
      ev = torch.cuda.Event()
      ev.record()
      self._pin_ev[k] = ev
    else:
      self.gamma[:n] = torch.tensor(vals, dtype=torch.float32)
    for i, (u, _, _, _) in enumerate(self.slots):
      if self._given is not None:
        g = self._given[i]
        if tuple(g.shape) != tuple(u.shape):
          raise ValueError('dropblock uniform has shape %s, expected %s' % (tuple(g.shape), tuple(u.shape)))
        u.copy_(g, non_blocking=True)
      else:
        u.uniform_(0.0, 1.0, generator=self._rng)

  def slot(self, shape, gamma_scale: float, H: int, W: int):
    """the walk reaches its next DropBlock call: (uniform buffer, gamma slot [1])"""
    i = self._i
    self._i += 1
    if self.known:
      if i >= len(self.slots) or tuple(self.slots[i][0].shape) != tuple(shape):
        raise RuntimeError('the DropBlock calls of this step differ from the ones the static buffers were made for')
      return self.slots[i][0], self.gamma[i:i + 1]
    if i >= self.CAP:
      raise RuntimeError('more than %d DropBlock calls in a step' % self.CAP)
    if self._given is not None:
      if i >= len(self._given) or tuple(self._given[i].shape) != tuple(shape):
        raise ValueError('dropblock uniform %d has the wrong shape (expected %s)' % (i, tuple(shape)))
      u = self._given[i].to(torch.float32).contiguous().clone()
    else:
      u = torch.rand(shape, generator=self._rng, device=self.device, dtype=torch.float32)
    self.slots.append((u, gamma_scale, H, W))
    self.gamma[i:i + 1].fill_(dropblock_gamma(gamma_scale, self._kp, H, W))
    return u, self.gamma[i:i + 1]

  def end(self):
    """after a step's walk"""
    if not self.known:
      self.known = True
    elif self._i != len(self.slots):
      raise RuntimeError('the step made %d DropBlock calls, %d were prepared' % (self._i, len(self.slots)))


def dropblock(ctx: Ctx, x: Var, keep_prob: float, gamma_scale: float, relu: bool) -> Var:
  """blocks.dropblock (nets/blocks.py:191-251) [+ the tf.nn.relu that follows it in the block].
  One Bernoulli seed mask per call, shared by the whole batch; draws come from ctx.next_uniform, or -- with
  ctx.db_static (DropBlockState) -- from that step's static buffers, with gamma read from device memory."""
  N, H, W, Cn = x.shape
  if ctx.dry:
    return Var(None, x.shape)
  bs = DROPBLOCK_SIZE
  st = getattr(ctx, 'db_static', None)
  if st is not None:
    u, gdev = st.slot((H - bs + 1, W - bs + 1, Cn), gamma_scale, H, W)
    keep, scale = ops.dropblock_mask(u, 0.0, H, W, Cn, bs, gamma_dev=gdev)
  else:
    gamma = dropblock_gamma(gamma_scale, keep_prob, H, W)
    u = ctx.next_uniform((H - bs + 1, W - bs + 1, Cn))
    keep, scale = ops.dropblock_mask(u, float(gamma), H, W, Cn, bs)
  y = Var(ops.dropblock_apply(x.data, keep, scale, relu=relu))
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, ops.dropblock_apply(y.grad, keep, scale, relu_mask_from=y.data if relu else None), True)
      y.grad = None
    ctx.record(bwd)
  return y


def relu_layer(ctx: Ctx, x: Var) -> Var:
  if ctx.dry:
    return Var(None, x.shape)
  y = Var(ops.relu_fwd(x.data))
  if ctx.tape is not None:
    def bwd():
      accum_grad(x, ops.relu_bwd(y.grad, y.data), True)
      y.grad = None
    ctx.record(bwd)
  return y
