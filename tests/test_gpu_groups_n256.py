"""GPU parity of whole conv -> batch-norm [-> + shortcut] -> ReLU groups, FORWARD AND BACKWARD, at the BASELINE shard size
(VERDICT round 3, item 6): batch 256, the 224 x 224-derived shapes of Assemble-ResNet-50 + D.

The teacher-forced whole-network harness (tests/model_parity.py) runs at batch 8-16 because whole-graph autograd at batch
256 does not fit the host; but ONE group does, and at batch 256 the product takes other paths than at batch 16: 256 x 256 /
8-wave tiles, the igemm3 kernel with two workgroups per CU, cost-model pixel splits and slab reduces in the weight
gradient, 1024-workgroup channel-sliced batch-norm reducers, the parity-class stride-2 input gradient, the weight-gradient
side streams.  Here each group runs through the product's own layer code (nn.conv_bn: fused statistics -> finalize ->
apply + packed ReLU mask; backward reduce -> finalize -> apply -> input gradient + weight gradient on the side streams)
with NOTHING forced, and is compared with the bf16-emulating oracle's autograd of the same group on the same inputs:

  forward output                      rel-L2 <= 4e-3   (the per-group bound of the teacher-forced forward harness)
  dx, dW, dgamma, dbeta               rel-L2 <= 6e-3   (the bound of the teacher-forced backward harness; 8-bit gradient
                                                        storage alone costs 2-3e-3: dy and dx are each rounded once)

The ReLU masks are the product's own here (no forcing), so a channel whose batch statistics differ in the last bit from the
oracle's flips a few mask bits: with >= 12 544 rows per channel the flipped fraction is ~1e-5 and stays inside the bounds
(measured values are printed on failure)."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
N = 256

# (name, k, cin, cout, H, stride, form)   form: 'relu' | 'linear' (no ReLU: the deferred / pre-add form) | 'merge' (+ identity
# shortcut before the ReLU) | 'proj' (block-final batch norm + projection-shortcut batch norm behind one ReLU)
GROUPS = [
    ('3x3 14x14x512->1024 (sk conv, stage 3)', 3, 512, 1024, 14, 1, 'relu'),
    ('3x3 28x28x128->256 (igemm3, 784 tiles)', 3, 128, 256, 28, 1, 'relu'),
    ('3x3 14x14x256->512 (256x256 tiles)', 3, 256, 512, 14, 1, 'relu'),
    ('3x3 7x7x512->1024 (stage 4)', 3, 512, 1024, 7, 1, 'relu'),
    ('3x3 7x7x256->512 (little branch)', 3, 256, 512, 7, 1, 'relu'),
    ('3x3 56x56x64->128 (stage 1 sk conv)', 3, 64, 128, 56, 1, 'relu'),
    ('3x3/2 112x112x64->64 (parity-class dgrad)', 3, 64, 64, 112, 2, 'relu'),
    ('3x3 112x112x32->64 (resident-halo kernels)', 3, 32, 64, 112, 1, 'relu'),
    ('1x1 56x56x256->64 (conv1)', 1, 256, 64, 56, 1, 'relu'),
    ('1x1 28x28x512->128 (conv1)', 1, 512, 128, 28, 1, 'relu'),
    ('1x1 14x14x512->1024 (conv3 + identity shortcut)', 1, 512, 1024, 14, 1, 'merge'),
    ('1x1 7x7x512->2048 (conv3 + projection shortcut 1024->2048)', 1, 512, 2048, 7, 1, 'proj'),
]


def _rel(a, b):
  return util.rel_l2(a.float().cpu(), b.float().cpu())


@pytest.mark.parametrize('group', GROUPS, ids=lambda g: g[0].split(' (')[0].replace(' ', '_'))
def test_group_forward_and_backward_at_batch_256(hip_lib, group):
  from assembled_cnn_amd import nn, ops
  from oracle import assembled_oracle as O
  name, k, cin, cout, H, stride, form = group
  dev = torch.device('cuda')
  torch.manual_seed(1234)
  import zlib
  g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0xffff)
  Ho = ops.out_size(H, k, stride)
  x = torch.randn((N, H, H, cin), generator=g).to(BF)
  dy = torch.randn((N, Ho, Ho, cout), generator=g).to(BF)
  sc_cin = 1024
  xs = torch.randn((N, Ho, Ho, sc_cin), generator=g).to(BF) if form == 'proj' else None        # shortcut branch input
  res = torch.randn((N, Ho, Ho, cout), generator=g).to(BF) if form == 'merge' else None

  # ---- oracle: its own conv2d_fixed_padding / batch_norm (bf16 storage emulation), autograd ----
  vs = O.VarStore(seed=7)
  oc = O.Ctx(vs, emulate_bf16=True)
  vs.begin_call()
  xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
  xsr = xs.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True) if xs is not None else None
  resr = res.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True) if res is not None else None
  rng = torch.Generator().manual_seed(99)

  def randomise_bn():   # gamma / beta away from (1, 0): a wrong coefficient must show
    for nme, t in vs.trainable.items():
      if nme.endswith('gamma') and float(t.detach().abs().sum()) == float(t.numel()):
        t.data.copy_(torch.rand(t.shape, generator=rng) + 0.5)
      elif nme.endswith('beta') and float(t.detach().abs().sum()) == 0.0:
        t.data.copy_(torch.randn(t.shape, generator=rng) * 0.2)

  def oracle_graph():
    shortcut = None
    if form == 'proj':
      shortcut = O.batch_norm(oc, O.conv2d_fixed_padding(oc, xsr, cout, 1, 1), True)
    elif form == 'merge':
      shortcut = resr
    y = O.conv2d_fixed_padding(oc, xr, cout, k, stride)
    return O.batch_norm(oc, y, True, relu=form != 'linear', residual=shortcut)

  with torch.no_grad():
    oracle_graph()            # creates the variables
  randomise_bn()
  vs.begin_call()
  zr = oracle_graph()
  leaves = [xr] + ([xsr] if xsr is not None else []) + ([resr] if resr is not None else []) + list(vs.trainable.values())
  grads = torch.autograd.grad(zr, leaves, dy.float().permute(0, 3, 1, 2))
  og = dict(zip(['x'] + (['xs'] if xsr is not None else []) + (['res'] if resr is not None else []) + list(vs.trainable.keys()),
                grads))

  # ---- product: the layer code of the training step, nothing forced ----
  arena = nn.ParamArena()
  dry = nn.Ctx(arena, True, True, 0.997, dev, False)
  if form == 'proj':
    conv_sc = nn.ConvKernel(dry, 1, sc_cin, cout)
    bn_sc = nn.BatchNorm(dry, cout)
  conv = nn.ConvKernel(dry, k, cin, cout)
  bn = nn.BatchNorm(dry, cout)
  arena.finalize(dev, 0)
  assert list(arena.specs.keys()) == list(vs.trainable.keys()), (list(arena.specs.keys()), list(vs.trainable.keys()))
  with torch.no_grad():
    for nme, t in vs.trainable.items():
      arena.w(nme).copy_(util.oracle_to_product_param(nme, t.detach().float()).to(dev))
  arena.refresh_shadows()
  arena.refresh_derived()
  arena.enable_side_stream()      # the product default: weight gradients beside the input-gradient chain
  ctx = nn.Ctx(arena, True, False, 0.997, dev, True)
  xv = nn.Var(x.to(dev))
  shortcut = None
  if form == 'proj':
    xsv = nn.Var(xs.to(dev))
    shortcut = nn.conv_bn(ctx, xsv, conv_sc, bn_sc, 1, relu=False)
  elif form == 'merge':
    shortcut = nn.Var(res.to(dev))
  out = nn.conv_bn(ctx, xv, conv, bn, stride, relu=form != 'linear', residual=shortcut, res_mode=1 if shortcut is not None else 0)
  z = out.data
  out.grad = dy.to(dev)
  ctx.backward()
  arena.join_side_stream()
  torch.cuda.synchronize()

  errs = []

  def check(what, got, ref, tol):
    r = _rel(got, ref)
    if not r <= tol:
      errs.append('%s: rel-L2 %.3e > %.1e' % (what, r, tol))

  check('forward', z, zr.detach().permute(0, 2, 3, 1), 4e-3)
  check('dx', xv.grad, og['x'].permute(0, 2, 3, 1), 6e-3)
  if form == 'proj':
    check('dx (shortcut branch)', xsv.grad, og['xs'].permute(0, 2, 3, 1), 6e-3)
  if form == 'merge':
    check('d shortcut', shortcut.grad, og['res'].permute(0, 2, 3, 1), 6e-3)
  for nme, t in vs.trainable.items():
    check(nme, util.product_to_oracle_grad(nme, arena.g(nme).cpu(), t), og[nme], 6e-3)
  assert not errs, '%s:\n  %s' % (name, '\n  '.join(errs))


# ---- whole bottleneck blocks at batch 256, nothing forced ---------------------------------------
# (name, H, cin, filters, blocks, stride): block_layer of Assemble-ResNet-50 + D (SK unit, sconv blur-pool k = 3): a projection
# block (ResNet-D shortcut: average pool -> 1x1 convolution -> batch norm, its gradient gathered in conv1's input-gradient
# epilogue; SK unit; blur-pool where the stride is 2; dual batch norm behind one ReLU; reordered backward tape) followed by
# an identity block (SK unit, conv3's batch norm + residual + ReLU, the lazily masked shortcut gradient as the fan-in
# addend of conv1's input gradient).  14 x 14 and 7 x 7 maps: the oracle's autograd at batch 256 fits the host.
BLOCKS = [
    ('stage 4: 14x14x1024 -> 7x7x2048, stride 2 (projection + blur-pool + identity)', 14, 1024, 512, 2, 2),
    ('stage 3 tail: 14x14x1024 -> 14x14x1024 (projection at stride 1 + identity)', 14, 1024, 256, 2, 1),
]
# What "nothing forced" can and cannot show.  Through ONE conv -> BN -> ReLU group (above) the product's own ReLU decisions
# differ from the oracle's on ~1e-5 of the elements.  Through a whole block the forward pass drifts by ~1e-2 (rel-L2: 5 - 7 bf16
# tensors in a row, the SK softmax, the blur-pool), activations within that distance of zero take the other ReLU branch, and a
# fraction p of flipped decisions moves every masked gradient behind them by ~sqrt(p): two bf16
# implementations of the same block that agree to 1e-2 in the forward pass agree to ~1e-1 in the gradients -- and so does the
# oracle with ITSELF: its bf16-emulating graph against its fp32 graph, measured here on the same inputs, is the noise floor the
# product is held against (x 1.5 + 1e-2), per tensor, next to a direction check (cosine >= 0.98).  The teacher-forced
# harness (tests/model_parity.py) is what checks the arithmetic of every layer at 6e-3 with the decisions pinned.
FLOOR_FACTOR, FLOOR_ABS, MIN_COS, FWD_TOL = 1.5, 1e-2, 0.98, 1.5e-2


@pytest.mark.parametrize('blk', BLOCKS, ids=lambda b: b[0].split(':')[0].replace(' ', '_'))
def test_sk_blocks_forward_and_backward_at_batch_256(hip_lib, blk):
  from assembled_cnn_amd import nn
  from assembled_cnn_amd.train import HParams
  from oracle import assembled_oracle as O
  name, H, cin, filters, nb, stride = blk
  dev = torch.device('cuda')
  import zlib
  g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0xffff)
  Ho = H // stride
  cout = 4 * filters
  x = torch.randn((N, H, H, cin), generator=g).abs_().to(BF)      # a block input is a ReLU output
  dy = (torch.randn((N, Ho, Ho, cout), generator=g) / (N * Ho * Ho) ** 0.5).to(BF)

  def oracle_run(emulate, values=None):
    vs = O.VarStore(seed=7)
    oc = O.Ctx(vs, emulate_bf16=emulate)
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)

    def graph():
      vs.begin_call()
      return O.block_layer(oc, xr, filters, True, O._bottleneck_block_v1, nb, stride, True, 'blk', zero_gamma=False,
                           use_resnet_d=True, use_sk_block=True, anti_alias_filter_size=3, anti_alias_type='sconv')
    with torch.no_grad():
      graph()            # creates the variables
    rng = torch.Generator().manual_seed(99)
    for nme, t in vs.trainable.items():      # gamma / beta away from (1, 0): a wrong coefficient must show
      if values is not None:
        t.data.copy_(values[nme])
      elif nme.endswith('gamma'):
        t.data.copy_(torch.rand(t.shape, generator=rng) * 0.5 + 0.5)
      elif nme.endswith('beta'):
        t.data.copy_(torch.randn(t.shape, generator=rng) * 0.1)
    z = graph()
    grads = torch.autograd.grad(z, [xr] + list(vs.trainable.values()), dy.float().permute(0, 3, 1, 2))
    return vs, z.detach(), dict(zip(['x'] + list(vs.trainable.keys()), grads))

  vs, zr, og = oracle_run(True)
  _, z32, og32 = oracle_run(False, {n_: t.detach().clone() for n_, t in vs.trainable.items()})

  hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True)
  m = hp.make_model(seed=0, device='cuda')
  dry = nn.Ctx(m.arena, True, True, 0.997, dev, False, m._layers)
  dry.keep_prob = 1.0
  m._block_layer(dry, nn.Var(None, (N, H, H, cin)), filters, nb, stride, 'blk', use_resnet_d=True)
  m.arena.finalize(dev, 0)
  assert list(m.arena.specs.keys()) == list(vs.trainable.keys()), (list(m.arena.specs.keys()), list(vs.trainable.keys()))
  with torch.no_grad():
    for nme, t in vs.trainable.items():
      m.arena.w(nme).copy_(util.oracle_to_product_param(nme, t.detach().float()).to(dev))
  m.arena.refresh_shadows()
  m.arena.refresh_derived()
  m.arena.enable_side_stream()
  ctx = nn.Ctx(m.arena, True, False, 0.997, dev, True, m._layers)
  ctx.keep_prob = 1.0
  xv = nn.Var(x.to(dev))
  out = m._block_layer(ctx, xv, filters, nb, stride, 'blk', use_resnet_d=True)
  z = out.data
  out.grad = dy.to(dev)
  m._ctx = ctx
  m.backward(None)            # the model's own backward driver (side streams, gradient-notification bookkeeping) over this tape
  torch.cuda.synchronize()

  rows, errs = [], []

  def cos(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))

  fwd = _rel(z, zr.permute(0, 2, 3, 1))
  fwd_floor = _rel(zr, z32)
  rows.append('%-58s %.3e   (oracle bf16 vs fp32: %.3e)' % ('forward', fwd, fwd_floor))
  if not fwd <= FWD_TOL:
    errs.append(rows[-1])

  def check(what, got, ref, ref32):
    r, floor, c = _rel(got, ref), _rel(ref, ref32), cos(got, ref)
    rows.append('%-58s %.3e   floor %.3e   cos %.4f' % (what, r, floor, c))
    if not (r <= FLOOR_FACTOR * floor + FLOOR_ABS and c >= MIN_COS):
      errs.append(rows[-1])

  check('dx', xv.grad, og['x'].permute(0, 2, 3, 1), og32['x'].permute(0, 2, 3, 1))
  for nme, t in vs.trainable.items():
    check(nme, util.product_to_oracle_grad(nme, m.arena.g(nme).cpu(), t), og[nme], og32[nme])
  print('\n'.join(rows))
  assert not errs, '%s: %d of %d outside the bounds:\n  %s' % (name, len(errs), len(rows), '\n  '.join(errs[:12]))
