"""GPU parity: the HBM-bound kernels (BN, pools, blur, SK/SE, loss, mixup, optimiser) through the C ABI
vs the CPU oracle on identical seeded bf16 inputs.

Tolerance: outputs are bf16 -> relative L2 <= 4e-3 and max |err| <= 2^-7 max|ref|; fp32 outputs
(statistics, parameter gradients, losses) relative <= 1e-4 unless noted."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0, shift=0.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(shape, generator=g) * scale + shift).to(BF)


def _close(out, ref, rel=4e-3, name=''):
  out = out.float().cpu()
  ref = ref.float()
  r = util.rel_l2(out, ref)
  m = util.max_abs(out, ref)
  lim = float(ref.abs().max()) * 2 ** -7 + 1e-6
  assert r <= rel, '%s rel_l2 %.3e' % (name, r)
  assert m <= lim, '%s max_abs %.3e > %.3e' % (name, m, lim)


def _nchw(t):
  return t.float().permute(0, 3, 1, 2)


def _nhwc(t):
  return t.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(4, 14, 14, 64), (2, 7, 7, 2048), (256, 1, 1, 32), (3, 9, 9, 72), (5, 6, 6, 1008)])
@pytest.mark.parametrize('relu,res_mode', [(True, 0), (False, 0), (True, 1), (False, 1), (True, 2)])
def test_bn_train_fwd_bwd(hip_lib, shape, relu, res_mode):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  N, H, W, Cn = shape
  if res_mode == 2 and (H % 2 or W % 2):
    pytest.skip('upsample residual needs even H, W')
  M = N * H * W
  x = _rand(shape, 1, scale=2.0, shift=0.3)
  gamma = torch.randn(Cn, generator=torch.Generator().manual_seed(2)) * 0.5 + 1
  beta = torch.randn(Cn, generator=torch.Generator().manual_seed(3)) * 0.2
  mm0, mv0 = torch.zeros(Cn), torch.ones(Cn)
  res = None
  if res_mode == 1:
    res = _rand(shape, 4)
  elif res_mode == 2:
    res = _rand((N, H // 2, W // 2, Cn), 4)
  dout = _rand(shape, 5)

  # oracle
  vs = O.VarStore(0)
  ctx = O.Ctx(vs, True)
  vs.begin_call()
  xr = _nchw(x).requires_grad_(True)
  rr = _nchw(res).requires_grad_(True) if res is not None else None
  g_, b_, _, _, mmn, mvn = vs.bn_vars(Cn, False, layer_name='bn')
  with torch.no_grad():
    g_.copy_(gamma)
    b_.copy_(beta)
  rfull = O.upsample2x_nearest(rr) if res_mode == 2 else rr
  yr = O.batch_norm(ctx, xr, True, momentum=0.9, relu=relu, residual=rfull, layer_name='bn')
  outs = [xr, g_, b_] + ([rr] if rr is not None else [])
  grads = torch.autograd.grad(yr, outs, _nchw(dout))

  # HIP
  xd = x.cuda()
  part = ops.bn_stats(xd, M, Cn)
  mm, mv = mm0.cuda(), mv0.cuda()
  gd, bd = gamma.cuda(), beta.cuda()
  mean, invstd, scale, shift = ops.bn_finalize(part, M, Cn, gd, bd, 1e-5, 0.9, mm, mv)
  y = ops.bn_apply(xd, M, Cn, scale, shift, res.cuda() if res is not None else None, res_mode, relu, H, W)
  _close(y, _nhwc(yr.detach()), name='bn fwd')
  xf = x.float().view(M, Cn)
  assert torch.allclose(mean.cpu(), xf.mean(0), rtol=1e-4, atol=1e-5)
  assert torch.allclose(invstd.cpu(), 1 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5), rtol=2e-4)
  assert torch.allclose(mm.cpu(), vs.pending_updates[mmn], rtol=1e-4, atol=1e-6), 'moving mean'
  assert torch.allclose(mv.cpu(), vs.pending_updates[mvn], rtol=2e-4, atol=1e-6), 'moving variance (Bessel)'

  dgamma = torch.empty(Cn, device='cuda')
  dbeta = torch.empty(Cn, device='cuda')
  want_dz = res is not None and relu
  dx, dz = ops.bn_bwd(dout.cuda(), xd, y, relu, M, Cn, gd, mean, invstd, dgamma, dbeta, want_dz)
  # the oracle mask comes from its own output; rows where the two outputs disagree in sign are
  # measure-zero for these sizes, so the tensors are comparable at bf16 tolerance
  _close(dx, _nhwc(grads[0]), rel=6e-3, name='bn dx')
  assert util.rel_l2(dgamma.cpu(), grads[1]) <= 2e-3
  assert util.rel_l2(dbeta.cpu(), grads[2]) <= 2e-3
  if relu:  # the packed 1-bit ReLU mask path must reproduce the bf16-output path bit for bit
    y2, mask = ops.bn_apply(xd, M, Cn, scale, shift, res.cuda() if res is not None else None, res_mode, True, H, W,
                            want_mask=True)
    assert torch.equal(y2, y) and mask.shape == (M, Cn // 8)
    bits = ((mask.cpu().to(torch.int32)[:, :, None] >> torch.arange(8, dtype=torch.int32)) & 1).view(M, Cn)
    assert torch.equal(bits.bool(), (y.float().cpu().view(M, Cn) > 0))
    dg2, db2 = torch.empty(Cn, device='cuda'), torch.empty(Cn, device='cuda')
    dx2, dz2 = ops.bn_bwd(dout.cuda(), xd, mask, True, M, Cn, gd, mean, invstd, dg2, db2, want_dz)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
    assert (dz2 is None and dz is None) or torch.equal(dz2, dz)
  if want_dz:
    dres_ref = _nhwc(grads[3])
    if res_mode == 2:
      _close(ops.upsample2x_bwd(dz), dres_ref, name='upsample bwd')
    else:
      _close(dz, dres_ref, name='bn dz')


def test_bn_infer(hip_lib):
  from assembled_cnn_amd import ops
  Cn, M = 128, 300
  x = _rand((M, Cn), 1)
  g = torch.rand(Cn) + 0.5
  b = torch.randn(Cn)
  mm = torch.randn(Cn) * 0.1
  mv = torch.rand(Cn) + 0.5
  scale, shift = ops.bn_infer_coeffs(Cn, g.cuda(), b.cuda(), mm.cuda(), mv.cuda(), 1e-5)
  y = ops.bn_apply(x.cuda(), M, Cn, scale, shift, None, 0, False)
  ref = (x.float() - mm) / torch.sqrt(mv + 1e-5) * g + b
  _close(y, ref, name='bn infer')


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(2, 112 // 4, 112 // 4, 64), (3, 15, 17, 8)])
def test_maxpool_same(hip_lib, shape):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  x = _rand(shape, 1)
  xr = _nchw(x).requires_grad_(True)
  yr = O.max_pool_same(xr, 3, 2)
  y, am = ops.maxpool3x3s2_fwd(x.cuda())
  assert torch.equal(y.float().cpu(), _nhwc(yr.detach()))
  dy = _rand(tuple(y.shape), 2)
  (gx,) = torch.autograd.grad(yr, xr, _nchw(dy))
  dx = ops.maxpool3x3s2_bwd(dy.cuda(), am, shape)
  _close(dx, _nhwc(gx), name='maxpool bwd')


@pytest.mark.parametrize('k,stride,pad,cv', [(3, 2, 1, False), (2, 2, 0, False), (2, 1, 0, True)])
@pytest.mark.parametrize('shape', [(2, 14, 14, 64), (2, 7, 7, 16)])
def test_avgpool(hip_lib, shape, k, stride, pad, cv):
  """the three shortcut poolings: BL 3x3/2 (divisor 9), ResNet-D 2x2/2, ResNet-D stride-1 SAME (valid count)."""
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  x = _rand(shape, 1)
  xr = _nchw(x).requires_grad_(True)
  if cv:
    yr = O.avg_pool_same(xr, k, stride)
  else:
    yr = O.avg_pool_valid(O.fixed_padding(xr, k), k, stride)
  Ho, Wo = yr.shape[2], yr.shape[3]
  y = ops.avgpool_fwd(x.cuda(), k, stride, pad, Ho, Wo, cv)
  _close(y, _nhwc(yr.detach()), name='avgpool fwd')
  dy = _rand(tuple(y.shape), 2)
  (gx,) = torch.autograd.grad(yr, xr, _nchw(dy))
  dx = ops.avgpool_bwd(dy.cuda(), shape, k, stride, pad, cv)
  _close(dx, _nhwc(gx), name='avgpool bwd')
  # fused fan-in add, in place in the addend's buffer
  add = _rand(shape, 3)
  buf = add.cuda().clone()
  out = ops.avgpool_bwd(dy.cuda(), shape, k, stride, pad, cv, addend=buf)
  assert out.data_ptr() == buf.data_ptr()
  _close(out, _nhwc(gx) + add.float(), name='avgpool bwd + addend')


@pytest.mark.parametrize('k', [3, 5, 2])
@pytest.mark.parametrize('shape', [(2, 14, 14, 64), (1, 7, 9, 8), (3, 4, 6, 16), (2, 8, 4, 8)])
def test_blurpool(hip_lib, shape, k):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  x = _rand(shape, 1)
  xr = _nchw(x).requires_grad_(True)
  yr = O.anti_aliased_downsample(O.Ctx(O.VarStore(0), False), xr, filt_size=k, stride=2)
  y = ops.blurpool_fwd(x.cuda(), k, 2)
  _close(y, _nhwc(yr.detach()), name='blur fwd')
  dy = _rand(tuple(y.shape), 2)
  (gx,) = torch.autograd.grad(yr, xr, _nchw(dy))
  dx = ops.blurpool_bwd(dy.cuda(), shape, k, 2)
  _close(dx, _nhwc(gx), name='blur bwd')


def test_gap_and_upsample(hip_lib):
  from assembled_cnn_amd import ops
  x = _rand((3, 7, 7, 2048), 1)
  y = ops.gap_fwd(x.cuda())
  _close(y.view(3, 2048), x.float().mean((1, 2)), name='gap')
  dy = _rand((3, 1, 1, 2048), 2)
  dx = ops.gap_bwd(dy.cuda(), (3, 7, 7, 2048))
  _close(dx, (dy.float() / 49).expand(3, 7, 7, 2048), name='gap bwd')
  g = _rand((2, 8, 8, 16), 3)
  _close(ops.upsample2x_bwd(g.cuda()), g.float().view(2, 4, 2, 4, 2, 16).sum((2, 4)), name='upsample bwd')
  # the lazily masked form: block sums of g * [mask bit] == block sums of the materialised masked gradient, bit for bit
  gg = _rand((3, 14, 14, 72), 4).cuda()
  mask = torch.randint(0, 256, (3 * 14 * 14, 9), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).cuda()
  assert torch.equal(ops.upsample2x_bwd(gg, mask), ops.upsample2x_bwd(ops.mask_apply(gg, mask)))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('N,H,W,F_', [(4, 14, 14, 64), (3, 7, 7, 512), (2, 5, 5, 32), (2, 56, 56, 64), (2, 6, 6, 24)])
def test_sk_select(hip_lib, N, H, W, F_):
  from assembled_cnn_amd import ops
  f = _rand((N, H, W, 2 * F_), 1).clamp(min=0)
  att = torch.randn((N, 2 * F_), generator=torch.Generator().manual_seed(2))
  dv = _rand((N, H, W, F_), 3)
  fr = f.float().requires_grad_(True)
  ar = att.clone().requires_grad_(True)
  a = torch.softmax(torch.stack([ar[:, :F_], ar[:, F_:]], 0), 0)
  v = fr[..., :F_] * a[0][:, None, None, :] + fr[..., F_:] * a[1][:, None, None, :]
  s_ref = (fr[..., :F_] + fr[..., F_:]).mean((1, 2))
  gf, ga = torch.autograd.grad(v, [fr, ar], dv.float())
  fd, ad = f.cuda(), att.cuda()
  _close(ops.sk_gap(fd, F_).view(N, F_), s_ref.detach(), name='sk gap')
  _close(ops.sk_select_fwd(fd, ad, F_), v.detach(), name='sk select')
  datt = ops.sk_select_bwd_att(fd, dv.cuda(), ad, F_)
  _close(datt.view(N, 2 * F_), ga, rel=6e-3, name='sk datt')
  ds = _rand((N, 1, 1, F_), 4)
  df = ops.sk_select_bwd_f(dv.cuda(), ad, ds.cuda(), F_)
  ref = gf + (ds.float() / (H * W)).repeat(1, 1, 1, 2).expand(N, H, W, 2 * F_)
  _close(df, ref, name='sk df')


def test_se_scale(hip_lib):
  from assembled_cnn_amd import ops
  N, H, W, Cn = 3, 7, 7, 256
  x = _rand((N, H, W, Cn), 1)
  e = torch.randn((N, Cn), generator=torch.Generator().manual_seed(2))
  dy = _rand((N, H, W, Cn), 3)
  xr = x.float().requires_grad_(True)
  er = e.clone().requires_grad_(True)
  y = xr * torch.sigmoid(er)[:, None, None, :]
  gx, ge = torch.autograd.grad(y, [xr, er], dy.float())
  _close(ops.se_scale_fwd(x.cuda(), e.cuda()), y.detach(), name='se fwd')
  _close(ops.se_scale_bwd_e(x.cuda(), dy.cuda(), e.cuda()).view(N, Cn), ge, rel=6e-3, name='se de')
  dsq = _rand((N, 1, 1, Cn), 4)
  ref = gx + (dsq.float() / (H * W)).expand(N, H, W, Cn)
  _close(ops.se_scale_bwd_x(dy.cuda(), e.cuda(), dsq.cuda()), ref, name='se dx')


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('eps,T', [(0.0, 0.0), (0.1, 0.0), (0.1, 1.0), (0.0, 4.0)])
def test_softmax_ce(hip_lib, eps, T):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  B, Cn, ld = 33, 1001, 1008
  g = torch.Generator().manual_seed(1)
  logits = torch.zeros(B, ld)
  logits[:, :Cn] = torch.randn(B, Cn, generator=g) * 3
  lam = torch.rand(B, 1, generator=g)
  y = lam * F.one_hot(torch.randint(0, Cn, (B,), generator=g), Cn) + (1 - lam) * F.one_hot(
      torch.randint(0, Cn, (B,), generator=g), Cn)
  teacher = torch.softmax(torch.randn(B, Cn, generator=g) * 3 / max(T, 1.0), 1) if T > 0 else None
  z = logits[:, :Cn].clone().requires_grad_(True)
  loss = O.softmax_cross_entropy(z, y, eps)
  if T > 0:
    loss = loss + O.kd_loss(z, teacher, T)
  (gz,) = torch.autograd.grad(loss * 128.0, z)
  rows, dz = ops.softmax_ce(logits.cuda(), ld, y.float().cuda(), teacher.cuda() if T > 0 else None, B, Cn, eps, T,
                            128.0, ld)
  assert abs(float(rows.mean()) - float(loss)) <= 1e-5 * abs(float(loss)) + 1e-5
  assert abs(float(ops.mean_f32(rows)) - float(loss)) <= 1e-5 * abs(float(loss)) + 1e-5
  dz = dz.view(B, ld).float().cpu()
  assert float(dz[:, Cn:].abs().max()) == 0.0, 'padding columns must be zero'
  _close(dz[:, :Cn], gz, name='dlogits')


def test_onehot_softmax_rows(hip_lib):
  from assembled_cnn_amd import ops
  lab = torch.tensor([0, 5, 1000, 7], dtype=torch.int32)
  oh = ops.onehot(lab.cuda(), 4, 1001).cpu()
  assert torch.equal(oh, F.one_hot(lab.long(), 1001).float())
  x = torch.randn(5, 1001)
  sm = ops.softmax_rows(x.cuda(), 5, 1001, 0.5).cpu()
  assert torch.allclose(sm, torch.softmax(x * 0.5, 1), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('mixup_type', [0, 1, 2])
@pytest.mark.parametrize('u8', [True, False])
def test_mixup_meansub(hip_lib, mixup_type, u8):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  Bin, H, W = 6, 10, 12
  img = util.seeded_images(Bin, H, W, 1)
  lam1 = torch.tensor([0.2, 0.9, 0.5])
  lam2 = torch.tensor([0.7, 0.1, 0.35])
  x = O.mean_image_subtraction(img.float())
  y = F.one_hot(torch.tensor([1, 2, 3, 4, 5, 6]), 11).float()
  if mixup_type == 0:
    xr, yr = x, y
  else:
    xr, yr, _ = O.mixup(x, y, lam1, keep_batch_size=(mixup_type == 2), lam2=lam2)
  src = img.cuda() if u8 else img.float().cuda()
  out = ops.mixup_meansub(src, mixup_type, lam1.cuda(), lam2.cuda())
  Bout = xr.shape[0]
  assert out.shape == (Bout, H + 6, W + 6, 4)
  _close(out[:, 3:3 + H, 3:3 + W, :3], xr, name='mixed images')
  halo = out.clone()
  halo[:, 3:3 + H, 3:3 + W, :3] = 0
  assert int((halo.view(torch.int16) != 0).sum()) == 0
  yo = ops.mixup_labels(y.cuda(), mixup_type, lam1.cuda(), lam2.cuda()).cpu()
  assert torch.allclose(yo, yr, atol=1e-6)
  if mixup_type:  # lambda = 1 is the identity on the first half (SURVEY 8c pin)
    one = torch.ones(3)
    out1 = ops.mixup_meansub(src, 1, one.cuda(), None)
    _close(out1[:, 3:3 + H, 3:3 + W, :3], x[:3], name='lambda=1')


def test_sgd_momentum(hip_lib):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  n = 1000 * 8 + 5
  g = torch.Generator().manual_seed(0)
  w = torch.randn(n, generator=g)
  a = torch.randn(n, generator=g) * 0.1
  gr = torch.randn(n, generator=g) * 128.0
  wd, ad, gd = w.cuda(), a.cuda(), gr.cuda()
  wb = torch.empty(n, dtype=BF, device='cuda')
  ops.sgd_momentum(wd, ad, gd, wb, 0.1, 0.9, 1e-4, 1.0 / 128.0)
  wr, ar = w.clone(), a.clone()
  O.momentum_step([wr], [gr / 128.0 + 1e-4 * w], [ar], 0.1, 0.9)
  assert torch.allclose(wd.cpu(), wr, rtol=1e-5, atol=1e-6)
  assert torch.allclose(ad.cpu(), ar, rtol=1e-5, atol=1e-6)
  assert torch.equal(wb.cpu(), wd.cpu().to(BF))


def test_elementwise(hip_lib):
  from assembled_cnn_amd import ops
  a, b = _rand((1024 * 8,), 1), _rand((1024 * 8,), 2)
  assert torch.equal(ops.add_bf16(a.cuda(), b.cuda()).cpu(), (a.float() + b.float()).to(BF))
  y = ops.relu_fwd(a.cuda())
  assert torch.equal(y.cpu(), a.clamp(min=0))
  assert torch.equal(ops.relu_bwd(b.cuda(), y).cpu(), torch.where(a > 0, b, torch.zeros_like(b)))
  z = torch.randn(7, 16)
  bias = torch.randn(10)
  zd = z.cuda()
  ops.bias_add_f32(zd, bias.cuda(), 7, 10, 16)
  ref = z.clone()
  ref[:, :10] += bias
  assert torch.allclose(zd.cpu(), ref)
  dz = _rand((7, 16), 3)
  db = torch.empty(10, device='cuda')
  ops.bias_grad_bf16(dz.cuda(), 7, 10, 16, db)
  assert torch.allclose(db.cpu(), dz.float()[:, :10].sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('M,Cn,relu', [(256, 32, True), (256, 256, True), (37, 72, False), (4096, 64, True)])
def test_bn_small_fused(hip_lib, M, Cn, relu):
  """one-launch BN of the [N,1,1,d] squeeze layers == the general three-kernel path == oracle batch_norm."""
  from assembled_cnn_amd import ops
  x = _rand((M, Cn), 1, scale=2.0)
  gamma = (torch.rand(Cn, generator=torch.Generator().manual_seed(2)) + 0.5).cuda()
  beta = (torch.randn(Cn, generator=torch.Generator().manual_seed(3)) * 0.1).cuda()
  mm, mv = torch.zeros(Cn).cuda(), torch.ones(Cn).cuda()
  y, mask, mean, invstd = ops.bn_small_fwd(x.cuda(), M, Cn, gamma, beta, 1e-5, 0.997, mm, mv, relu, True)
  # general path on the same input
  mm2, mv2 = torch.zeros(Cn).cuda(), torch.ones(Cn).cuda()
  part = ops.bn_stats(x.cuda(), M, Cn)
  mean2, invstd2, scale, shift = ops.bn_finalize(part, M, Cn, gamma, beta, 1e-5, 0.997, mm2, mv2)
  if relu:
    y2, mask2 = ops.bn_apply(x.cuda(), M, Cn, scale, shift, relu=True, want_mask=True)
    assert torch.equal(mask, mask2)
  else:
    y2 = ops.bn_apply(x.cuda(), M, Cn, scale, shift, relu=False)
    assert mask is None
  assert torch.allclose(mean, mean2, rtol=1e-5, atol=1e-6) and torch.allclose(invstd, invstd2, rtol=1e-5)
  assert torch.allclose(mm, mm2, rtol=1e-5, atol=1e-7) and torch.allclose(mv, mv2, rtol=1e-5)
  _close(y, y2.float().cpu(), name='bn_small fwd vs general')
  # fp32 reference
  xf = x.float()
  mu, var = xf.mean(0), xf.var(0, unbiased=False)
  ref = (xf - mu) / torch.sqrt(var + 1e-5) * gamma.cpu() + beta.cpu()
  _close(y, ref.clamp(min=0) if relu else ref, name='bn_small fwd vs fp32')
  # backward
  dy = _rand((M, Cn), 4)
  dg, db = torch.empty(Cn).cuda(), torch.empty(Cn).cuda()
  dx = ops.bn_small_bwd(dy.cuda(), x.cuda(), mask if relu else None, M, Cn, gamma, mean, invstd, dg, db)
  dg2, db2 = torch.empty(Cn).cuda(), torch.empty(Cn).cuda()
  dx2, _ = ops.bn_bwd(dy.cuda(), x.cuda(), mask if relu else None, relu, M, Cn, gamma, mean2, invstd2, dg2, db2, False)
  _close(dx, dx2.float().cpu(), name='bn_small bwd vs general')
  assert torch.allclose(dg, dg2, rtol=1e-4, atol=1e-3) and torch.allclose(db, db2, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('N,H,W,F_', [(4, 14, 14, 64), (3, 28, 28, 32), (2, 7, 7, 1024), (5, 9, 9, 24), (2, 56, 56, 64)])
def test_sk_unit_with_bn_applied_on_the_fly_equals_materialised_path(hip_lib, N, H, W, F_):
  """csrc/sk_fused.hip against the kernels it replaces, on the same conv output y:
  bn_apply -> sk_gap / sk_select / sk_select_bwd_att / sk_select_bwd_f -> bn_bwd  ==  the on-the-fly forms.
  Forward pieces are bit-identical (f is rounded to bf16 at the same point); the BN backward differs only by the bf16
  rounding of df that the fused form skips."""
  from assembled_cnn_amd import ops
  C2, M, HW = 2 * F_, N * H * W, H * W
  g = torch.Generator().manual_seed(F_ + H)
  y = (torch.randn((N, H, W, C2), generator=g) * 1.5 + 0.2).to(torch.bfloat16).cuda()
  gamma = (torch.rand(C2, generator=g) + 0.5).cuda()
  beta = (torch.randn(C2, generator=g) * 0.3).cuda()
  att = (torch.randn((N, 1, 1, C2), generator=g) * 2).cuda()
  dv = torch.randn((N, H, W, F_), generator=g).to(torch.bfloat16).cuda()
  ds = torch.randn((N, 1, 1, F_), generator=g).to(torch.bfloat16).cuda()
  part = ops.bn_stats(y.view(M, C2), M, C2)
  mean, invstd, scale, shift = ops.bn_finalize(part, M, C2, gamma, beta, 1e-5, 0.997, None, None)
  # materialising path
  f, mask = ops.bn_apply(y.view(M, C2), M, C2, scale, shift, None, 0, True, H, W, want_mask=True)
  f = f.view(N, H, W, C2)
  s0 = ops.sk_gap(f, F_)
  v0 = ops.sk_select_fwd(f, att, F_)
  datt0 = ops.sk_select_bwd_att(f, dv, att, F_)
  df = ops.sk_select_bwd_f(dv, att, ds, F_)
  dg0, db0 = torch.empty(C2, device='cuda'), torch.empty(C2, device='cuda')
  dy0, _ = ops.bn_bwd(df.view(M, C2), y.view(M, C2), mask, True, M, C2, gamma, mean, invstd, dg0, db0, False)
  # on the fly
  s1 = ops.sk_gap_bn(y, scale, shift, F_)
  v1 = ops.sk_select_bn_fwd(y, scale, shift, att, F_)
  datt1 = ops.sk_select_bn_bwd_att(y, scale, shift, dv, att, F_)
  dg1, db1 = torch.empty(C2, device='cuda'), torch.empty(C2, device='cuda')
  dy1 = ops.sk_bn_bwd(dv, att, ds, y, scale, shift, gamma, mean, invstd, dg1, db1, F_)
  assert torch.equal(v0, v1), 'select'
  assert util.rel_l2(s1.float(), s0.float()) <= 2e-3 and util.rel_l2(datt1.float(), datt0.float()) <= 4e-3
  assert util.rel_l2(dg1, dg0) <= 4e-3 and util.rel_l2(db1, db0) <= 4e-3
  assert util.rel_l2(dy1.float(), dy0.float()) <= 6e-3
  # and against fp32 autograd of the definition (nets/blocks.py:126-152 + tf.layers.batch_normalization backward)
  yy = y.float().cpu().requires_grad_(True)
  mu = yy.mean((0, 1, 2))
  var = ((yy - mu) ** 2).mean((0, 1, 2))
  fr = torch.relu((yy - mu) * torch.rsqrt(var + 1e-5) * gamma.cpu() + beta.cpu())
  f0, f1 = fr[..., :F_], fr[..., F_:]
  a = torch.softmax(torch.stack([att.cpu()[..., :F_], att.cpu()[..., F_:]], 0), 0)
  sr = (f0 + f1).mean((1, 2), keepdim=True)
  vr = a[0] * f0 + a[1] * f1
  (gy,) = torch.autograd.grad([vr, sr], [yy], [dv.float().cpu(), ds.float().cpu()])
  assert util.rel_l2(v1.float().cpu(), vr.detach()) <= 4e-3
  assert util.rel_l2(s1.float().cpu(), sr.detach()) <= 4e-3
  assert util.rel_l2(dy1.float().cpu(), gy) <= 8e-3
  # factorised batch-norm reduction: per-image statistics out of the pooled-sum and gate-gradient passes + a tiny finalize
  # == the reduce pass over the whole tensor (same dz, sums regrouped per image)
  s2, mst = ops.sk_gap_bn(y, scale, shift, F_, mean, invstd)
  datt2, gst = ops.sk_select_bn_bwd_att(y, scale, shift, dv, att, F_, mean, invstd)
  assert torch.equal(s2, s1)
  assert util.rel_l2(datt2.float(), datt1.float()) <= 2e-3          # (f0 - f1) dV summed as f0 dV - f1 dV
  dg2, db2 = torch.empty(C2, device='cuda'), torch.empty(C2, device='cuda')
  dy2 = ops.sk_bn_bwd(dv, att, ds, y, scale, shift, gamma, mean, invstd, dg2, db2, F_, gst, mst)
  assert util.rel_l2(dg2, dg1) <= 1e-3 and util.rel_l2(db2, db1) <= 1e-3, (util.rel_l2(dg2, dg1), util.rel_l2(db2, db1))
  assert util.rel_l2(dy2.float(), dy1.float()) <= 2e-3
  assert util.rel_l2(dy2.float().cpu(), gy) <= 8e-3


# ---------------------------------------------------------------------------------------------------
# small dense layers (csrc/dense_small.hip): the [N,1,1,C] squeeze / excite / classifier layers
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,ldo,f32,addend', [
    (256, 512, 128, 512, True, False),      # sk_fc_2 forward (fp32 logits)
    (256, 128, 256, 128, False, False),     # sk_fc_1 input gradient
    (256, 1001, 2048, 1008, True, False),   # the classifier: N tail inside a quad, padded rows
    (256, 2048, 1008, 2048, False, True),   # its input gradient over the zero-padded 1008 logits, with an addend
    (7, 40, 48, 40, False, False),          # ragged rows / channel tile
    (33, 36, 16, 36, True, False),
])
def test_dense_small_vs_fp32(hip_lib, M, N, K, ldo, f32, addend):
  from assembled_cnn_amd.ops import L, _ptr, _stream, check
  p = _rand((M, K), 1)
  q = _rand((N, K), 2, scale=K ** -0.5)
  add = _rand((M, ldo), 3) if addend else None
  out = torch.full((M, ldo), 7.0, dtype=torch.float32 if f32 else BF).cuda()
  pc, qc = p.cuda(), q.cuda()
  addc = add.cuda() if addend else None
  check(L().asm_dense_small(_ptr(pc), K, _ptr(qc), K, M, N, K, _ptr(out), ldo, 1 if f32 else 0, _ptr(addc), _stream()), 'dense_small')
  ref = p.float() @ q.float().t()
  if addend:
    ref = ref + add.float()[:, :N]
  got = out.float().cpu()
  if f32:
    assert util.rel_l2(got[:, :N], ref) <= 2e-6, util.rel_l2(got[:, :N], ref)
  else:
    _close(got[:, :N], ref, name='dense_small')
  assert bool((got[:, N:] == 0.0).all()), 'pad columns N .. ldo-1 are written as zeros (include/asm_hip.h)'


def test_dense_small_refuses_a_row_stride_it_cannot_zero(hip_lib):
  """`out` owns whole rows and its pad columns are written as zeros by the tile that holds column N-1: a row stride beyond
  that tile (a column slice of a wider matrix, or > 31 pad columns) is an argument error, not a half-zeroed row"""
  from assembled_cnn_amd.ops import L, _ptr, _stream
  from assembled_cnn_amd import lib
  p, q = _rand((32, 64), 1).cuda(), _rand((32, 64), 2).cuda()
  out = torch.full((32, 64), 7.0, dtype=torch.float32).cuda()
  assert L().asm_dense_small(_ptr(p), 64, _ptr(q), 64, 32, 32, 64, _ptr(out), 64, 1, None, _stream()) == lib.ASM_EINVAL
  assert bool((out == 7.0).all()), 'nothing may be written by a refused call'
  assert L().asm_dense_small(_ptr(p), 64, _ptr(q), 64, 32, 30, 64, _ptr(out), 32, 1, None, _stream()) == lib.ASM_OK


def test_dense_layers_route_through_dense_small_and_match_the_conv_kernels(hip_lib, monkeypatch):
  """ops.conv_fprop / conv_dgrad of a [N,1,1,C] layer: the dense kernel == the implicit-GEMM convolution (A/B knob)"""
  from assembled_cnn_amd import ops
  N_, Cn, K = 256, 128, 512
  x = _rand((N_, 1, 1, Cn), 1).cuda()
  w = _rand((K, 1, 1, Cn), 2, scale=Cn ** -0.5).cuda()
  dy = _rand((N_, 1, 1, K), 3).cuda()
  wt = torch.zeros((Cn, 1, 1, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, 1, 1, Cn)
  outs = {}
  for knob in ('1', '0'):
    util.set_knob(monkeypatch, 'ASM_DENSE_SMALL', knob)
    d = ops.make_conv_desc(N_, 1, 1, Cn, K, 1, 1, 1, out_f32=True)
    y, _ = ops.conv_fprop(d, x, w, False)
    db = ops.make_conv_desc(N_, 1, 1, Cn, K, 1, 1, 1)
    yb, _ = ops.conv_fprop(db, x, w, False)
    dx = ops.conv_dgrad(db, dy, wt)
    outs[knob] = (y.float().cpu(), yb.float().cpu(), dx.float().cpu())
  assert util.rel_l2(outs['1'][0], outs['0'][0]) <= 2e-6
  _close(outs['1'][1], outs['0'][1], name='dense fprop bf16')
  _close(outs['1'][2], outs['0'][2], name='dense dgrad')


def test_sk_attention_path_fused_vs_unfused_whole_unit(hip_lib, monkeypatch):
  """One SK bottleneck network step with the squeeze layers on csrc/dense_small.hip vs on the convolution kernels:
  logits and every parameter gradient agree to bf16 noise."""
  from tests import model_parity as MP
  res = {}
  for knob in ('1', '0'):
    util.set_knob(monkeypatch, 'ASM_DENSE_SMALL', knob)
    om, pm = MP.make_pair('a-r50', 'cuda', 8, 64)
    _, x, _ = MP.inputs(8, 64)
    lp = pm(x.cuda(), True, use_resnet_d=False)
    dl = torch.zeros((8, 1, 1, pm.ldc), dtype=BF, device='cuda')
    dl[:, 0, 0, :1001] = (torch.softmax(lp.float(), 1) / 8).to(BF)
    pm.backward(dl)
    torch.cuda.synchronize()
    grads = {n: pm.arena.g(n).float().cpu().clone() for n in pm.arena.specs}
    res[knob] = (lp.float().cpu().clone(), grads)
  assert util.rel_l2(res['1'][0], res['0'][0]) <= 6e-2
  ga, gb = res['1'][1], res['0'][1]
  worst = 1.0
  for n in ga:
    if 'sk_block' in n and float(gb[n].norm()) > 0:
      cos = float((ga[n] * gb[n]).sum() / (ga[n].norm() * gb[n].norm() + 1e-30))
      worst = min(worst, cos)
      assert cos >= 0.9, '%s: gradient cosine %.3f between the two squeeze-layer paths' % (n, cos)
  fa = torch.cat([g.reshape(-1) for g in ga.values()])
  fb = torch.cat([g.reshape(-1) for g in gb.values()])
  assert float((fa * fb).sum() / (fa.norm() * fb.norm())) >= 0.95


@pytest.mark.parametrize('M,Cin,Cout,ldy', [(256, 256, 1024, 1024), (256, 2048, 1001, 1008), (37, 40, 24, 24), (128, 64, 32, 32)])
def test_dense_small_wgrad_vs_fp32_and_conv_kernel(hip_lib, M, Cin, Cout, ldy, monkeypatch):
  from assembled_cnn_amd import ops
  x = _rand((M, 1, 1, Cin), 1).cuda()
  dy = torch.zeros((M, 1, 1, ldy), dtype=BF)
  dy[..., :Cout] = _rand((M, 1, 1, Cout), 2)
  dy = dy.cuda()
  d = ops.make_conv_desc(M, 1, 1, Cin, Cout, 1, 1, 1, ldy=ldy if ldy != Cout else 0)
  ref = dy.float().cpu().view(M, ldy)[:, :Cout].t() @ x.float().cpu().view(M, Cin)
  got = {}
  for knob in ('1', '0'):
    util.set_knob(monkeypatch, 'ASM_DENSE_SMALL', knob)
    dw = torch.full((Cout, 1, 1, Cin), 3.0, device='cuda')
    ops.conv_wgrad(d, x, dy, dw)
    got[knob] = dw.cpu().view(Cout, Cin)
  assert util.rel_l2(got['1'], ref) <= 2e-6, util.rel_l2(got['1'], ref)
  assert util.rel_l2(got['0'], ref) <= 1e-4


@pytest.mark.parametrize('M,Cn', [(2 * 28 * 28, 256), (1000, 72), (256 * 7 * 7, 2048), (3 * 56 * 56, 128)])
def test_bn_bwd_dual_equals_two_separate_backwards(hip_lib, M, Cn):
  """out = relu(bn_a(xa) + bn_b(xb)) (block-final + projection-shortcut batch norm): one reduce + one apply for both,
  against the oracle's autograd through its two batch norms, and == two asm_bn_bwd_reduce / finalize / apply chains on the
  same (dout, mask)."""
  from assembled_cnn_amd import ops
  g = torch.Generator(device='cuda').manual_seed(5)
  xa = torch.randn((M, Cn), generator=g, device='cuda').to(BF)
  xb = (torch.randn((M, Cn), generator=g, device='cuda') * 2 + 0.3).to(BF)
  dy = torch.randn((M, Cn), generator=g, device='cuda').to(BF)
  mask = torch.randint(0, 256, (M, Cn // 8), generator=g, device='cuda', dtype=torch.uint8)
  bns = []
  for x in (xa, xb):
    gamma = torch.rand(Cn, generator=g, device='cuda') + 0.5
    xf = x.float()
    mean = xf.mean(0)
    invstd = 1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)
    bns.append((gamma, mean.contiguous(), invstd.contiguous()))
  outs = [torch.empty(Cn, device='cuda') for _ in range(4)]
  dxa, dxb = ops.bn_bwd_dual(dy, xa, xb, mask, M, Cn, bns[0] + (outs[0], outs[1]), bns[1] + (outs[2], outs[3]))
  # against the oracle: autograd through its two training-mode batch norms, fed the masked gradient g = dy * [mask bit]
  from oracle import assembled_oracle as O
  bits = ((mask.cpu().to(torch.int32)[:, :, None] >> torch.arange(8, dtype=torch.int32)) & 1).reshape(M, Cn).float()
  vs = O.VarStore(0)
  octx = O.Ctx(vs, True)
  vs.begin_call()
  leaves, total = [], 0.0
  for tag, x, (gamma, _, _) in (('a', xa, bns[0]), ('b', xb, bns[1])):
    xr = x.float().cpu().t().reshape(1, Cn, M, 1).clone().requires_grad_(True)      # NCHW with H = M, W = 1
    g_, b_ = vs.bn_vars(Cn, False, layer_name='bn_' + tag)[:2]
    with torch.no_grad():
      g_.copy_(gamma.cpu())
    total = total + O.batch_norm(octx, xr, True, momentum=0.9, layer_name='bn_' + tag)
    leaves += [xr, g_, b_]
  gr = torch.autograd.grad(total, leaves, (dy.float().cpu() * bits).t().reshape(1, Cn, M, 1))
  for i, (dx, dg, db) in enumerate(((dxa, outs[0], outs[1]), (dxb, outs[2], outs[3]))):
    _close(dx, gr[3 * i].reshape(Cn, M).t(), rel=6e-3, name='dual dx vs oracle')
    assert util.rel_l2(dg.cpu(), gr[3 * i + 1]) <= 2e-3 and util.rel_l2(db.cpu(), gr[3 * i + 2]) <= 2e-3
  for x, (gamma, mean, invstd), dx, dg, db in ((xa, bns[0], dxa, outs[0], outs[1]), (xb, bns[1], dxb, outs[2], outs[3])):
    dg2, db2 = torch.empty(Cn, device='cuda'), torch.empty(Cn, device='cuda')
    dx2, _ = ops.bn_bwd(dy, x, mask, True, M, Cn, gamma, mean, invstd, dg2, db2, False)
    _close(dx, dx2.float().cpu(), rel=1e-3, name='dual dx')
    assert torch.allclose(dg, dg2, rtol=1e-4, atol=1e-3) and torch.allclose(db, db2, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('M,Cn,relu', [(2 * 28 * 28, 256, True), (1000, 72, True), (3 * 56 * 56, 128, False)])
def test_bn_apply_dual_is_bit_identical_to_two_passes(hip_lib, M, Cn, relu):
  """[relu](bn_a(xa) + bf16(bn_b(xb))) in one pass == asm_bn_apply(xb) followed by asm_bn_apply(xa, residual)"""
  from assembled_cnn_amd import ops
  g = torch.Generator(device='cuda').manual_seed(9)
  xa = torch.randn((M, Cn), generator=g, device='cuda').to(BF)
  xb = (torch.randn((M, Cn), generator=g, device='cuda') * 2 + 0.3).to(BF)
  co = [torch.randn(Cn, generator=g, device='cuda') * 0.5 + (1.0 if i % 2 == 0 else 0.0) for i in range(4)]
  zb = ops.bn_apply(xb, M, Cn, co[2], co[3], relu=False)
  if relu:
    y1, m1 = ops.bn_apply(xa, M, Cn, co[0], co[1], zb, 1, True, want_mask=True)
    y2, m2 = ops.bn_apply_dual(xa, xb, M, Cn, co[0], co[1], co[2], co[3], True, want_mask=True)
    assert torch.equal(m1, m2)
  else:
    y1 = ops.bn_apply(xa, M, Cn, co[0], co[1], zb, 1, False)
    y2 = ops.bn_apply_dual(xa, xb, M, Cn, co[0], co[1], co[2], co[3], False)
  assert torch.equal(y1, y2)
  # and against the definition in fp32 on the host: [relu](scale_a * xa + shift_a + bf16(scale_b * xb + shift_b))
  c = [t.cpu() for t in co]
  want = xa.float().cpu() * c[0] + c[1] + (xb.float().cpu() * c[2] + c[3]).to(BF).float()
  _close(y2, torch.relu(want) if relu else want, name='dual apply vs definition')
