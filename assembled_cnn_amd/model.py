"""Assemble-ResNet topology walker over the HIP layers -- the drop-in for the reference's model class.

Mirrors the operator interface of the reference for this path:

  * ``Model(resnet_size, data_format=None, num_classes=None, resnet_version=1, dtype=..., no_downsample,
    zero_gamma, use_se_block, use_sk_block, bn_momentum, embedding_size, anti_alias_filter_size,
    anti_alias_type, pool_type, loss_type, bl_alpha, bl_beta)``        functions/model_fns.py:141-157
  * ``model(inputs NHWC, training, reuse=False, use_resnet_d=False, keep_prob=1.0,
    return_embedding=False) -> logits [B, num_classes]``               nets/resnet_model.py:305-310
  * the same errors: ValueError for a bad resnet_version / resnet_size / dtype, NotImplementedError for
    non-bottleneck sizes and unknown pool / loss types.            nets/resnet_model.py:200-215,
                                                                    functions/model_fns.py:131-135

Variables are created in the reference's creation order with TF-style names
(``resnet_model/.../conv2d_N/kernel`` ...), but kernels are stored KRSC (see ``hwio_to_krsc``).
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict
from typing import Optional

import torch

from . import nn, ops

# the two torch stream primitives the walker uses, as module attributes so that a CPU test can stand in for them
_current_stream = torch.cuda.current_stream
_stream_ctx = torch.cuda.stream
from .nn import BatchNorm, ConvKernel, Ctx, SEUnit, SKUnit, Var, conv_bn


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


ALLOWED_DTYPES = ('bf16',)          # compute dtype of the HIP path
KNOWN_DTYPES = ('bf16', 'fp16', 'fp32')


class Model(object):
  def __init__(self, resnet_size, data_format=None, num_classes=None, resnet_version=1, dtype='bf16',
               no_downsample=False, zero_gamma=False, use_se_block=False, use_sk_block=False,
               bn_momentum=0.997, embedding_size=0, anti_alias_filter_size=0, anti_alias_type="",
               pool_type='gap', loss_type='softmax', bl_alpha=2, bl_beta=4, seed=0, device='cuda'):
    if resnet_version not in (1, 2):
      raise ValueError('Resnet version should be 1 or 2. See README for citations.')
    if int(resnet_size) < 50:
      raise NotImplementedError('only bottleneck ResNets (resnet_size >= 50) are implemented')
    if dtype not in KNOWN_DTYPES:
      raise ValueError('dtype must be one of: {}'.format(KNOWN_DTYPES))
    if dtype not in ALLOWED_DTYPES:
      raise NotImplementedError('the MI355X path computes in bf16 (fp32 master weights); got dtype=%s' % dtype)
    if data_format not in (None, 'channels_last'):
      raise NotImplementedError('the MI355X path is NHWC (channels_last) only')
    if pool_type not in ('gap', 'gem', 'flatten'):
      raise NotImplementedError
    if loss_type == 'softmax':
      self.dense_bias_init = 0.0
    elif loss_type in ('sigmoid', 'focal', 'anchor'):
      self.dense_bias_init = -math.log(num_classes - 1)
    else:
      raise NotImplementedError
    self.resnet_size = int(resnet_size)
    self.resnet_version = resnet_version
    self.num_classes = num_classes
    self.num_filters = 64
    self.kernel_size = 7
    self.conv_stride = 2
    self.first_pool_size = 3
    self.first_pool_stride = 2
    self.block_sizes = get_block_sizes(self.resnet_size, resnet_version)
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
    self.alpha = bl_alpha
    self.beta = bl_beta
    self.dtype = dtype
    self.seed = seed
    self.device = torch.device(device)
    self.arena = nn.ParamArena()
    self._layers = []
    self._built_with_d: Optional[bool] = None
    self.taps: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    self.ldc = nn._round_up(num_classes, 8) if num_classes else 0
    self._ctx: Optional[Ctx] = None
    self._db_rng = None
    self._bl_stream = None
    # side streams of THIS model: None = as the ASM_WGRAD_STREAM / ASM_BL_STREAMS switches of the process say (default on),
    # True / False = set for this model only (Trainer.set_streams / calibrate_streams)
    self.side_streams: Optional[bool] = None

  # -----------------------------------------------------------------------------------------------
  def build(self, input_hw=(224, 224), use_resnet_d=False, batch=2):
    """Create all variables (shape-only walk), allocate and initialise the arenas."""
    if self.arena.finalized:
      return
    ctx = Ctx(self.arena, True, True, self.bn_momentum, self.device, False, self._layers)
    ctx.keep_prob = 1.0
    x = Var(None, (batch, input_hw[0] + 6, input_hw[1] + 6, 4), needs_grad=False)
    self._walk(ctx, x, use_resnet_d, False)
    self._built_with_d = use_resnet_d
    self.arena.finalize(self.device, self.seed)
    # Weight gradients are leaves of the backward graph, so they run on a second HIP stream beside the
    # dgrad -> BN-backward chain and fill the tail rounds of the 1-workgroup-per-CU convolution tiles: -0.4 .. -0.8 % step
    # time in same-box A/B runs (round 1: 8330 vs 8190 img/s; round 2: 29.59 / 29.71 vs 29.83 / 29.84 ms).
    # ASM_WGRAD_STREAM=0 keeps everything on the compute stream.  dp.GradSync joins these streams before every bucket launch.
    if self._streams_on('ASM_WGRAD_STREAM'):
      self.arena.enable_side_stream()     # no-op on the CPU test double

  def _streams_on(self, knob: str) -> bool:
    return self.side_streams if self.side_streams is not None else ops.knob(knob, '1') != '0'

  def __call__(self, inputs, training, reuse=False, use_resnet_d=False, keep_prob=1.0, return_embedding=False,
               record_tape=None, prepadded=False, dropblock_uniforms=None, db_static=None):
    """inputs: [N, H, W, 3] float32 / bfloat16 NHWC (already mean-subtracted), or with ``prepadded`` the
    zero-haloed [N, H+6, W+6, 4] bf16 buffer produced by ops.mixup_meansub.  Returns float32 logits
    [N, num_classes] (a view of the padded logits buffer)."""
    keep_prob = float(keep_prob)
    if not 0.0 < keep_prob <= 1.0:   # nets/blocks.py:218-220
      raise ValueError('keep_prob must be a scalar tensor or a float in the range (0, 1], got %g' % keep_prob)
    if prepadded:
      xp = inputs
      hw = (inputs.shape[1] - 6, inputs.shape[2] - 6)
    else:
      hw = (inputs.shape[1], inputs.shape[2])
    if not self.arena.finalized:
      self.build(hw, use_resnet_d)
    if use_resnet_d != self._built_with_d:
      raise ValueError('model variables were created with use_resnet_d=%s' % self._built_with_d)
    if not prepadded:
      xp = ops.stem_pad_input(inputs.contiguous())
    tape = training if record_tape is None else record_tape
    ctx = Ctx(self.arena, training, False, self.bn_momentum, self.device, tape, self._layers)
    # blocks.dropblock is the identity unless training with keep_prob < 1 (nets/blocks.py:208-213)
    ctx.keep_prob = keep_prob if (training and keep_prob < 1.0) else 1.0
    # db_static (nn.DropBlockState, prepared by the caller for this step): every DropBlock call of the topology runs, whatever
    # keep_prob is -- at keep_prob = 1 its gamma is 0, the mask all ones and the scale exactly 1, i.e. the identity the
    # reference returns early for (nets/blocks.py:208-213; the un-fused block tail rounds to bf16 twice more than the fused
    # one the eager identity path takes) -- so that the step's launch sequence is static
    ctx.db_static = db_static if training else None
    ctx.uniforms = iter(dropblock_uniforms) if (dropblock_uniforms is not None and ctx.db_static is None) else None
    if ctx.keep_prob < 1.0 and ctx.uniforms is None and ctx.db_static is None:
      if self._db_rng is None:
        self._db_rng = torch.Generator(device=self.device)
        self._db_rng.manual_seed(self.seed + 12345)
      ctx.rng = self._db_rng
    out = self._walk(ctx, Var(xp, needs_grad=False), use_resnet_d, return_embedding)
    self.taps = ctx.taps
    self._ctx = ctx
    return out

  def backward(self, dlogits: torch.Tensor):
    """Run the recorded tape.  dlogits: bf16 [N, 1, 1, ldc] (d loss / d logits, zero padded)."""
    if self._ctx is None or self._ctx.tape is None:
      raise RuntimeError('no forward pass with a tape to differentiate')
    self._ctx.dlogits = dlogits
    self.arena.release_grads()      # a previous backward that raised mid-block must not leave notifications queued
    self.arena.defer_grads(False)
    self.arena._deferred = []
    cuda = self.arena.side_stream is not None
    self.arena.compute_stream = torch.cuda.current_stream() if cuda else None    # looked up once, not per weight gradient
    try:
      self._ctx.backward()
    finally:
      self.arena.compute_stream = None
    self._ctx = None
    self.arena.join_side_stream()

  # -----------------------------------------------------------------------------------------------
  def _branch_stream(self, ctx: Ctx, x: Var):
    """the HIP stream the big branch of a BigLittle stage runs on (forward, and blocks 2..n of its backward), or None"""
    if ctx.dry or x.data is None or not x.data.is_cuda or not self._streams_on('ASM_BL_STREAMS'):
      return None
    if getattr(ctx, 'keep_prob', 1.0) < 1.0 and getattr(ctx, 'db_static', None) is None:
      return None          # DropBlock draws come from one generator in creation order (static buffers are drawn beforehand)
    if self._bl_stream is None:
      self._bl_stream = torch.cuda.Stream(device=x.data.device)
      self.arena.extra_streams.append(self._bl_stream)
    return self._bl_stream

  def _bl_backward(self, ctx: Ctx, side, big_out: Var, big_first, big_rest, little):
    """Backward of the two branches of a BigLittle stage as ONE tape entry: the big branch's blocks 2..n (half resolution,
    small-M deep-K tiles) on the branch stream beside the little branch (full resolution, bandwidth-bound) on the compute
    stream -- the same pairing as in the forward pass; then, streams joined, the big branch's first block, whose input
    gradient carries the fan-in add with the little branch's.  The host feeds the two streams in turns of a few tape
    entries.  Gradient-ready notifications of the big branch are deferred until the little branch (created later) is
    through: dp.GradSync's watermark stays monotone.  ASM_BL_BWD=0 (or ASM_BL_STREAMS=0) keeps the plain tape order."""
    arena = ctx.arena
    TURN = 3        # tape entries per turn, about one bottleneck block (1 / 3 / 6 / all at once measured the same)

    def run():
      main = _current_stream()
      big, lit = list(reversed(big_rest)), list(reversed(little))
      # The big branch's output gradient (whatever lazy form it is in) was allocated on the compute stream and is read on
      # the branch stream; its consumer drops it while the little branch keeps allocating on the compute stream, and the
      # caching allocator would hand the block out again while the branch stream still reads it.  Kept alive until the join.
      keep = (big_out._grad, big_out.grad_mask, big_out.pre_dy, big_out.pool_grad)
      ops.stream_join(side, main)       # the merge's backward produced both branches' output gradients on the compute stream
      bi = li = 0
      try:
        while bi < len(big) or li < len(lit):
          if bi < len(big):
            arena.defer_grads(True)
            arena.compute_stream = side
            with _stream_ctx(side):
              for fn in big[bi:bi + TURN]:
                fn()
            bi += TURN
            arena.compute_stream = main
            arena.defer_grads(False)
          for fn in lit[li:li + TURN]:
            fn()
          li += TURN
      finally:
        arena.compute_stream = main
        arena.defer_grads(False)
      ops.stream_join(main, side)
      del keep
      arena.flush_deferred()
      for fn in reversed(big_first):
        fn()
    return run

  def _bottleneck(self, ctx: Ctx, x: Var, filters, projection, strides, zero_gamma, aa_size, aa_type,
                  last_relu=True, expansion=4, db_gamma_scale=None) -> Var:
    """_bottleneck_block_v1 (nets/resnet_model.py:35-97).  db_gamma_scale: DropBlock gamma multiplier of this
    stage (0.25 / 1.0 for stages 3 / 4, :434-453) or None; active only while training with keep_prob < 1."""
    L = ctx.layer
    cin = x.shape[3]
    db = db_gamma_scale if (db_gamma_scale is not None and (getattr(ctx, 'keep_prob', 1.0) < 1.0 or
                                                           getattr(ctx, 'db_static', None) is not None)) else None
    kp = getattr(ctx, 'keep_prob', 1.0)
    shortcut = x
    t0 = len(ctx.tape) if ctx.tape is not None else None
    if projection is not None:
      shortcut = projection(x)
      if db is not None:
        shortcut = nn.dropblock(ctx, shortcut, kp, db, relu=False)                      # :46-47
    t1 = len(ctx.tape) if ctx.tape is not None else None
    c1, b1 = L(lambda: ConvKernel(ctx, 1, cin, filters)), L(lambda: BatchNorm(ctx, filters))
    if db is None:
      h = conv_bn(ctx, x, c1, b1, 1, relu=True)
    else:
      h = nn.dropblock(ctx, conv_bn(ctx, x, c1, b1, 1, relu=False), kp, db, relu=True)   # BN -> dropblock -> relu
    s3 = 1 if 'sconv' in aa_type else strides
    if self.use_sk_block:
      h = L(lambda: SKUnit(ctx, filters, filters))(ctx, h, s3)
      if db is not None:
        h = nn.dropblock(ctx, h, kp, db, relu=False)                                     # :62-63
    else:
      c2, b2 = L(lambda: ConvKernel(ctx, 3, filters, filters)), L(lambda: BatchNorm(ctx, filters))
      if db is None:
        h = conv_bn(ctx, h, c2, b2, s3, relu=True)
      else:
        h = nn.dropblock(ctx, conv_bn(ctx, h, c2, b2, s3, relu=False), kp, db, relu=True)
    if 'sconv' in aa_type and strides != 1:
      h = nn.blur_pool(ctx, h, aa_size, strides)
    if projection is not None and t0 is not None and t1 > t0:
      # Backward order of a projection block: block-final layer -> SHORTCUT branch -> main branch (the forward order, and
      # with it the variable creation order, stays shortcut first).  The shortcut's average pool then leaves its gradient
      # in pooled form and conv1's input gradient gathers it in its epilogue (nn.Var.pool_grad) instead of a scatter
      # pass over the full-resolution block input followed by an add.  The shortcut's variables were created before the
      # main branch's, so their gradient-ready notifications are held until the main branch is through (dp.GradSync).
      tape, arena = ctx.tape, ctx.arena
      t2 = len(tape)
      tape[t0:t2] = [arena.release_grads] + tape[t1:t2] + [arena.pass_grads] + tape[t0:t1] + [arena.hold_grads]
    cout = expansion * filters
    conv3 = L(lambda: ConvKernel(ctx, 1, filters, cout))
    bn3 = L(lambda: BatchNorm(ctx, cout, zero_gamma=zero_gamma))
    se = L(lambda: SEUnit(ctx, cout)) if self.use_se_block else None
    if se is not None or db is not None:
      h = conv_bn(ctx, h, conv3, bn3, 1, relu=False)
      if db is not None:
        h = nn.dropblock(ctx, h, kp, db, relu=False)                                     # :86-87
      if se is not None:
        h = se(ctx, h)
      return self._add_relu(ctx, h, shortcut, last_relu)
    return conv_bn(ctx, h, conv3, bn3, 1, relu=last_relu, residual=shortcut, res_mode=1)

  @staticmethod
  def _add_relu(ctx: Ctx, a: Var, b: Var, relu: bool) -> Var:
    if ctx.dry:
      return Var(None, a.shape)
    s = ops.add_bf16(a.data, b.data)
    out = Var(ops.relu_fwd(s) if relu else s)
    if ctx.tape is not None:
      def bwd():
        g = ops.relu_bwd(out.grad, out.data) if relu else out.grad
        nn.accum_grad(a, g, relu)
        nn.accum_grad(b, g, False)
        out.grad = None
      ctx.record(bwd)
    return out

  def _block_layer(self, ctx: Ctx, x: Var, filters, num_blocks, strides, name, use_resnet_d=False,
                   use_bl=False, last_relu=True, expansion=4, db_gamma_scale=None) -> Var:
    """block_layer (nets/resnet_model.py:99-163)."""
    L = ctx.layer
    filters_out = filters * expansion
    aa_size, aa_type = self.anti_alias_filter_size, self.anti_alias_type

    def shortcut_conv_bn(inp: Var, stride: int) -> Var:
      cin = inp.shape[3]
      return conv_bn(ctx, inp, L(lambda: ConvKernel(ctx, 1, cin, filters_out)),
                     L(lambda: BatchNorm(ctx, filters_out)), stride, relu=False)

    def projection_shortcut(inp):      # :107-121
      if 'proj' in aa_type and strides != 1:
        return shortcut_conv_bn(nn.blur_pool(ctx, inp, aa_size, strides), 1)
      return shortcut_conv_bn(inp, strides)

    def resnet_d_projection_shortcut(inp):  # :123-131
      if strides > 1:
        inp = nn.avg_pool(ctx, inp, 2, strides, 0, False)
      else:
        inp = nn.avg_pool(ctx, inp, 2, 1, 0, True)
      return shortcut_conv_bn(inp, 1)

    def bl_projection_shortcut(inp):   # :133-141
      if strides > 1:
        inp = nn.avg_pool(ctx, inp, 3, strides, 1, False)
      return shortcut_conv_bn(inp, 1)

    if use_resnet_d:
      proj = resnet_d_projection_shortcut
    elif use_bl:
      proj = bl_projection_shortcut
    else:
      proj = projection_shortcut

    # first block: projection + stride + anti-alias args; last_relu is NOT forwarded (:151-155)
    x = self._bottleneck(ctx, x, filters, proj, strides, self.zero_gamma, aa_size, aa_type, True, expansion,
                         db_gamma_scale)
    ctx.first_block_end = len(ctx.tape) if ctx.tape is not None else None     # (the BigLittle backward split, _bl_backward)
    for i in range(1, num_blocks):     # :157-161
      x = self._bottleneck(ctx, x, filters, None, 1, self.zero_gamma, 0, "",
                           last_relu if i == num_blocks - 1 else True, expansion, db_gamma_scale)
    ctx.tap(name, x)
    return x

  # -----------------------------------------------------------------------------------------------
  def _walk(self, ctx: Ctx, x: Var, use_resnet_d: bool, return_embedding: bool):
    """Model.__call__ (nets/resnet_model.py:305-599)."""
    L = ctx.layer
    nf = self.num_filters
    v2 = self.resnet_version == 2

    # ---- stem ------------------------------------------------------------------------------------
    if use_resnet_d:                                      # :328-358
      if v2:
        ctx.push_scope('stage0')
      c1 = L(lambda: ConvKernel(ctx, 3, 3, nf // 2, stem=True))
      b1 = L(lambda: BatchNorm(ctx, nf // 2))
      c2 = L(lambda: ConvKernel(ctx, 3, nf // 2, nf // 2))
      b2 = L(lambda: BatchNorm(ctx, nf // 2))
      c3 = L(lambda: ConvKernel(ctx, 3, nf // 2, nf))
      if v2:
        ctx.pop_scope()
        ctx.push_scope('stage0')
      b3 = L(lambda: BatchNorm(ctx, nf))
      if v2:
        ctx.pop_scope()
      x = conv_bn(ctx, x, c1, b1, self.conv_stride, relu=True)
      x = conv_bn(ctx, x, c2, b2, 1, relu=True)
      x = conv_bn(ctx, x, c3, b3, 1, relu=True, tap_pre='initial_conv')
    else:                                                 # :359-381
      if v2:
        ctx.push_scope('stage0')
      c1 = L(lambda: ConvKernel(ctx, self.kernel_size, 3, nf, stem=True))
      if v2:
        ctx.pop_scope()
        ctx.push_scope('stage0')
      b1 = L(lambda: BatchNorm(ctx, nf))
      if v2:
        ctx.pop_scope()
      x = conv_bn(ctx, x, c1, b1, self.conv_stride, relu=True, tap_pre='initial_conv')

    if self.first_pool_size:
      if v2:                                              # blModule0 :384-419
        ctx.push_scope('stage0/pool')
        a = self.alpha
        cb = L(lambda: ConvKernel(ctx, 3, nf, nf)); bb = L(lambda: BatchNorm(ctx, nf))
        big0 = conv_bn(ctx, x, cb, bb, 2, relu=False)
        cl1 = L(lambda: ConvKernel(ctx, 3, nf, nf // a)); bl1 = L(lambda: BatchNorm(ctx, nf // a))
        l0 = conv_bn(ctx, x, cl1, bl1, 1, relu=True)
        cl2 = L(lambda: ConvKernel(ctx, 3, nf // a, nf // a)); bl2 = L(lambda: BatchNorm(ctx, nf // a))
        l0 = conv_bn(ctx, l0, cl2, bl2, 2, relu=True)
        cl3 = L(lambda: ConvKernel(ctx, 1, nf // a, nf)); bl3 = L(lambda: BatchNorm(ctx, nf))
        x = conv_bn(ctx, l0, cl3, bl3, 1, relu=True, residual=big0, res_mode=1)   # relu(big0 + little0) :413
        cm = L(lambda: ConvKernel(ctx, 1, nf, nf)); bm = L(lambda: BatchNorm(ctx, nf))
        x = conv_bn(ctx, x, cm, bm, 1, relu=True)
        ctx.pop_scope()
      else:                                               # :420-425
        x = nn.max_pool_3x3_s2_same(ctx, x)
        ctx.tap('initial_max_pool', x)

    # ---- stages ----------------------------------------------------------------------------------
    for i, num_blocks in enumerate(self.block_sizes):
      num_filters = nf * (2 ** i)
      dbs = {2: 0.25, 3: 1.0}.get(i)                       # dropblock_for_group3 / 4 (:434-453)
      if v2 and i < 3:                                    # :455-516
        ctx.push_scope('stage{}'.format(i + 1))
        ctx.push_scope('big{}'.format(i + 1))
        # The big branch (half resolution: small-M, deep-K tiles that leave CUs idle) and the little branch (full
        # resolution, bandwidth-bound) are independent until the merge: the big branch is enqueued on a second HIP stream
        # so the two fill each other's gaps -- here in the forward pass, and again in the backward pass (_bl_backward).
        # Both directions are fenced by stream waits, so every cross-stream tensor is produced before it is read and the
        # caching allocator only ever recycles a block inside the stream that owns it.  ASM_BL_STREAMS=0 keeps one stream.
        side = self._branch_stream(ctx, x)
        tb0 = len(ctx.tape) if ctx.tape is not None else None
        if side is not None:
          ops.stream_join(side, _current_stream())
          with _stream_ctx(side):
            big = self._block_layer(ctx, x, num_filters, num_blocks - 1, 2, 'big{}'.format(i + 1),
                                    use_bl=True, last_relu=False, db_gamma_scale=dbs)
        else:
          big = self._block_layer(ctx, x, num_filters, num_blocks - 1, 2, 'big{}'.format(i + 1),
                                  use_bl=True, last_relu=False, db_gamma_scale=dbs)
        tb1, tbf = (len(ctx.tape), ctx.first_block_end) if ctx.tape is not None else (None, None)
        ctx.pop_scope()
        ctx.push_scope('little{}'.format(i + 1))
        little = self._block_layer(ctx, x, num_filters // self.alpha, max(1, num_blocks // self.beta - 1), 1,
                                   'little{}'.format(i + 1), use_bl=True, db_gamma_scale=dbs)
        cin_l = little.shape[3]
        ce = L(lambda: ConvKernel(ctx, 1, cin_l, num_filters * 4))
        be = L(lambda: BatchNorm(ctx, num_filters * 4))
        ctx.pop_scope()
        if side is not None:
          ops.stream_join(_current_stream(), side)
          if tb0 is not None and tb1 > tbf and ops.knob('ASM_BL_BWD', '1') != '0':
            tape = ctx.tape
            tape[tb0:] = [self._bl_backward(ctx, side, big, tape[tb0:tbf], tape[tbf:tb1], tape[tb1:])]
        # relu(BN(little_e) + UpSampling2D(big)) :493-501
        x = conv_bn(ctx, little, ce, be, 1, relu=True, residual=big, res_mode=2)
        ctx.push_scope('merge{}'.format(i + 1))
        x = self._block_layer(ctx, x, num_filters, 1, self.block_strides[i], 'merge{}'.format(i + 1), use_bl=True,
                              db_gamma_scale=dbs)
        ctx.pop_scope()
        ctx.pop_scope()
      elif v2 and i == 3:                                 # :518-534
        ctx.push_scope('stage{}'.format(i + 1))
        x = self._block_layer(ctx, x, num_filters, num_blocks, self.block_strides[i],
                              'block_layer{}'.format(i + 1), use_resnet_d=use_resnet_d, use_bl=True,
                              db_gamma_scale=dbs)
        ctx.pop_scope()
      else:                                               # :536-549
        x = self._block_layer(ctx, x, num_filters, num_blocks, self.block_strides[i],
                              'block_layer{}'.format(i + 1), use_resnet_d=use_resnet_d, db_gamma_scale=dbs)

    # ---- head :555-599 ---------------------------------------------------------------------------
    if self.pool_type == 'gap':                           # :560-569
      x = nn.global_avg_pool(ctx, x)
    elif self.pool_type == 'gem':
      x = nn.gem_pool(ctx, x)
    else:
      x = nn.flatten_pool(ctx, x)
    ctx.tap('final_reduce_mean', x)
    if self.embedding_size > 0:
      cin_e = x.shape[3]
      ce = L(lambda: ConvKernel(ctx, 1, cin_e, self.embedding_size, layer_name='embedding_dense'))
      be = L(lambda: BatchNorm(ctx, self.embedding_size, layer_name='embedding_dense_batch_normalization'))
      if return_embedding:
        x = conv_bn(ctx, x, ce, be, 1, relu=False)
      else:
        x = conv_bn(ctx, x, ce, be, 1, relu=True)   # relu after squeeze (:591-592)
    if return_embedding:
      return None if ctx.dry else x.data.view(x.shape[0], x.shape[3]).float()
    return self._dense(ctx, x)

  def _dense(self, ctx: Ctx, x: Var):
    """tf.layers.dense (nets/resnet_model.py:595-597): kernel stored [units][in], bias added in fp32."""
    L = ctx.layer
    cin = x.shape[3]
    nc = self.num_classes
    dense = L(lambda: ConvKernel(ctx, 1, cin, nc, dense=True))
    if ctx.dry:
      bias_name = dense.name[:-len('kernel')] + 'bias'
      ctx.arena.register(bias_name, (nc,), True, nn.const_init((nc,), self.dense_bias_init))
      self._bias_name = bias_name
      return None
    a = ctx.arena
    logits, dense_bwd, _ = nn.conv_plain(ctx, x, dense, out_f32=True, ldy=self.ldc)
    N = x.shape[0]
    ops.bias_add_f32(logits, a.w(self._bias_name), N, nc, self.ldc)
    ctx.taps['final_dense'] = logits
    if ctx.tape is not None:
      def bwd():
        dz = ctx.dlogits
        if dz is None:
          raise RuntimeError('backward() needs d(loss)/d(logits)')
        ops.bias_grad_bf16(dz, N, nc, self.ldc, a.g(self._bias_name))
        dense_bwd(dz)
      ctx.record(bwd)
    self.logits_padded = logits
    return logits.view(N, self.ldc)[:, :nc]

  # -----------------------------------------------------------------------------------------------
  def trainable_variables(self) -> "OrderedDict[str, torch.Tensor]":
    """name -> fp32 master view, in creation order (tf.trainable_variables())."""
    return OrderedDict((n, self.arena.w(n)) for n in self.arena.specs)

  def num_params(self) -> int:
    return self.arena.num_params()


def hwio_to_krsc(w_hwio: torch.Tensor) -> torch.Tensor:
  """TF conv kernel [k, k, Cin, Cout] -> this package's [Cout, k, k, Cin]."""
  return w_hwio.permute(3, 0, 1, 2).contiguous()


def krsc_to_hwio(w_krsc: torch.Tensor) -> torch.Tensor:
  return w_krsc.permute(1, 2, 3, 0).contiguous()
