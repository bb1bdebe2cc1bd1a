"""GPU parity: MFMA convolution kernels (fprop / dgrad / wgrad, fused BN statistics, stem path) called
through the C ABI vs the CPU oracle on the same seeded, bf16-rounded inputs.

Tolerances (bf16 in, fp32 accumulate, bf16 out): relative L2 <= 4e-3 (~2^-8) per tensor and
max |err| <= 2^-7 * max|ref| (one bf16 ulp of the largest value + accumulation-order slack);
fp32 outputs (wgrad, f32 fprop): relative L2 <= 2e-3 (K-order only; inputs are identical bf16)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(shape, generator=g) * scale).to(BF)


def _ref_conv(x_nhwc, w_krsc, stride):
  """oracle conv2d_fixed_padding on bf16-rounded inputs, fp32 math."""
  from oracle import assembled_oracle as O
  k = w_krsc.shape[1]
  w_hwio = w_krsc.float().permute(1, 2, 3, 0)
  y = O._conv_raw(x_nhwc.float().permute(0, 3, 1, 2), w_hwio, k, stride)
  return y.permute(0, 2, 3, 1).contiguous()


def _check(out, ref, rel=4e-3, name=''):
  out = out.float().cpu()
  r = util.rel_l2(out, ref)
  m = util.max_abs(out, ref)
  lim = float(ref.abs().max()) * 2 ** -7 + 1e-6
  assert r <= rel, '%s rel_l2 %.3e > %.1e' % (name, r, rel)
  assert m <= lim, '%s max_abs %.3e > %.3e' % (name, m, lim)


SHAPES = [
    # N, H,  W,  C,   K,   k, stride
    (2, 16, 16, 64, 64, 3, 1),      # BN=64, BK=64 (igemm variants); conv_halo_kernel Ci 64 -> 64 in the default mode
    (2, 16, 16, 64, 128, 3, 1),     # BN=128 (SK conv shape, small)
    (3, 7, 7, 256, 512, 3, 1),      # M=147 (ragged M tile), 4 N-tiles
    (2, 14, 14, 128, 128, 3, 2),    # strided 3x3
    (2, 16, 16, 64, 256, 1, 1),     # 1x1
    (2, 16, 16, 256, 64, 1, 1),     # 1x1 reduce
    (2, 14, 14, 256, 512, 1, 2),    # strided 1x1 projection
    (2, 12, 12, 32, 64, 3, 1),      # BK=32
    (2, 12, 12, 32, 32, 3, 2),      # BN=64 with K=32 (masked N), BK=32, strided
    (2, 9, 9, 72, 40, 3, 1),        # channel tail (72 % 32 = 8), K not multiple of 64
    (256, 1, 1, 64, 32, 1, 1),      # SK fc1 shape
    (4, 15, 15, 64, 64, 3, 2),      # odd spatial, strided
    (2, 16, 32, 64, 32, 3, 1),      # conv_halo_kernel: 8 x 16 patches with a resident halo, Ci 64 -> 32
    (3, 8, 16, 32, 64, 3, 1),       # conv_halo_kernel: one patch per image, Ci 32 -> 64
    (1, 24, 48, 32, 32, 3, 1),      # conv_halo_kernel: 3 x 3 patches, Ci 32 -> 32
    (2, 14, 14, 128, 256, 3, 1),    # igemm3_kernel (rows resident across the taps): 2 chunks, 4 row tiles with a ragged last one
    (5, 7, 7, 192, 136, 3, 1),      # igemm3_kernel: 3 chunks, K tail inside the second N tile (136 = 128 + 8), images inside a tile
    (1, 9, 30, 128, 128, 3, 1),     # igemm3_kernel: widest map it takes (W = 30), H != W
    (3, 5, 11, 256, 72, 3, 1),      # igemm3_kernel: narrow N tile (72 of 128), 4 chunks, odd H / W
]


@pytest.mark.parametrize('mode', [0, 1], ids=['per-layer-kernels', 'general-kernel'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_fprop_dgrad_wgrad_vs_oracle(hip_lib, shape, mode, monkeypatch):
  from assembled_cnn_amd import ops
  util.set_knob(monkeypatch, 'ASM_IGEMM_MODE', str(mode))   # 1: every layer on igemm_kernel (the general form)
  util.set_knob(monkeypatch, 'ASM_DENSE_SMALL', '0')        # [N,1,1,C] shapes too: this test is about the convolution kernels
  N, H, W, Cn, K, k, stride = shape
  x = _rand((N, H, W, Cn), 1)
  w = _rand((K, k, k, Cn), 2, scale=(1.0 / (k * k * Cn)) ** 0.5)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  xd, wd = x.cuda(), w.cuda()

  # forward (+ fused statistics)
  y, stats = ops.conv_fprop(d, xd, wd, want_stats=True)
  ref = _ref_conv(x, w, stride)
  assert tuple(y.shape) == tuple(ref.shape)
  _check(y, ref, name='fprop')
  yb = y.float()
  s = stats.sum(0).cpu()
  assert torch.allclose(s[0], yb.sum((0, 1, 2)).cpu(), rtol=1e-4, atol=1e-2), 'fused sum'
  assert torch.allclose(s[1], (yb * yb).sum((0, 1, 2)).cpu(), rtol=1e-4, atol=1e-2), 'fused sum of squares'
  # same launch without statistics gives the identical tensor
  y2, _ = ops.conv_fprop(d, xd, wd, want_stats=False)
  assert torch.equal(y, y2)

  # backward: reference via autograd on the oracle conv
  dy = _rand(tuple(ref.shape), 3)
  xr = x.float().requires_grad_(True)
  wr = w.float().requires_grad_(True)
  from oracle import assembled_oracle as O
  yr = O._conv_raw(xr.permute(0, 3, 1, 2), wr.permute(1, 2, 3, 0), k, stride)
  gx, gw = torch.autograd.grad(yr, [xr, wr], dy.float().permute(0, 3, 1, 2))

  wt = torch.zeros((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(wd, wt, K, k, k, Cn)
  assert torch.equal(wt.cpu(), w.permute(3, 1, 2, 0).contiguous())
  dx = ops.conv_dgrad(d, dy.cuda(), wt)
  _check(dx, gx, name='dgrad')
  # fused fan-in add: dgrad(dy) + addend, with addend = dgrad(dy) itself -> exactly 2x in bf16
  dx2 = ops.conv_dgrad(d, dy.cuda(), wt, addend=dx)
  assert torch.equal(dx2.float(), dx.float() * 2)

  dw = torch.empty((K, k, k, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, xd, dy.cuda(), dw)
  r = util.rel_l2(dw.cpu(), gw)
  assert r <= 2e-3, 'wgrad rel_l2 %.3e' % r


def test_fprop_f32_out_with_padded_ld(hip_lib):
  """dense-like: K=100 (not a multiple of 8), fp32 output with row stride 104, then the padded backward."""
  from assembled_cnn_amd import ops
  N, Cn, K, ld = 64, 256, 100, 104
  x = _rand((N, 1, 1, Cn), 5)
  w = _rand((K, 1, 1, Cn), 6, scale=Cn ** -0.5)
  d = ops.make_conv_desc(N, 1, 1, Cn, K, 1, 1, 1, ldy=ld, out_f32=True)
  y, _ = ops.conv_fprop(d, x.cuda(), w.cuda())
  ref = x.float().view(N, Cn) @ w.float().view(K, Cn).t()
  assert util.rel_l2(y.view(N, ld)[:, :K].cpu(), ref) <= 2e-3
  # backward with dy padded to ld columns (zeros)
  dy = torch.zeros((N, 1, 1, ld), dtype=BF)
  dy[..., :K] = _rand((N, 1, 1, K), 7)
  dd = ops.make_conv_desc(N, 1, 1, Cn, K, 1, 1, 1, ldy=ld)
  dw = torch.empty((K, 1, 1, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(dd, x.cuda(), dy.cuda(), dw)
  gw = dy.float().view(N, ld)[:, :K].t() @ x.float().view(N, Cn)
  assert util.rel_l2(dw.view(K, Cn).cpu(), gw) <= 2e-3
  wt = torch.zeros((Cn, 1, 1, ld), dtype=BF, device='cuda')
  ops.filter_transpose(w.cuda(), wt, K, 1, 1, Cn, ld)
  d2 = ops.make_conv_desc(N, 1, 1, Cn, ld, 1, 1, 1)
  dx = ops.conv_dgrad(d2, dy.cuda(), wt)
  gx = dy.float().view(N, ld)[:, :K] @ w.float().view(K, Cn)
  _check(dx.view(N, Cn), gx, name='dense dgrad')


@pytest.mark.parametrize('ksize,cout', [(7, 64), (3, 32)])
def test_stem_conv_path(hip_lib, ksize, cout):
  """3-channel first conv (7x7/2 and the ResNet-D 3x3/2) through the halo-buffer packing."""
  from assembled_cnn_amd import nn, ops
  N, H, W = 3, 32, 32
  dev = torch.device('cuda')
  arena = nn.ParamArena()
  c = nn.Ctx(arena, True, True, 0.997, dev, False)
  conv = nn.ConvKernel(c, ksize, 3, cout, stem=True)
  arena.finalize(dev, 0)
  x = _rand((N, H, W, 3), 11, scale=50.0)
  xp = ops.stem_pad_input(x.cuda())
  assert xp.shape == (N, H + 6, W + 6, 4)
  assert torch.equal(xp[:, 3:3 + H, 3:3 + W, :3].cpu(), x)
  halo = xp.clone()
  halo[:, 3:3 + H, 3:3 + W, :3] = 0
  assert int((halo.view(torch.int16) != 0).sum()) == 0, 'halo / 4th channel must be zero'
  d = conv.desc(N, H, W, 2)
  y, _ = conv.fprop(d, xp, False)
  w = arena.wb(conv.name).cpu()
  ref = _ref_conv(x, w, 2)
  _check(y, ref, name='stem fprop')
  dy = _rand(tuple(ref.shape), 12)
  conv.backward(d, xp, dy.cuda(), False)
  xr = x.float()
  wr = w.float().requires_grad_(True)
  from oracle import assembled_oracle as O
  yr = O._conv_raw(xr.permute(0, 3, 1, 2), wr.permute(1, 2, 3, 0), ksize, 2)
  (gw,) = torch.autograd.grad(yr, [wr], dy.float().permute(0, 3, 1, 2))
  assert util.rel_l2(arena.g(conv.name).cpu(), gw) <= 2e-3


def test_mfma_vs_naive_at_full_size(hip_lib):
  """BASELINE-size layer (batch 64 slice of the 256x56x56x64 -> 128 SK conv): the MFMA kernels against
  the one-thread-per-output debug kernels, where the CPU oracle would take minutes."""
  from assembled_cnn_amd import ops
  L = ops.L()
  N, H, W, Cn, K, k = 64, 56, 56, 64, 128, 3
  g = torch.Generator(device='cuda').manual_seed(0)
  x = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, k, k, Cn), generator=g, device='cuda') * (k * k * Cn) ** -0.5).to(BF)
  dy = torch.randn((N, H, W, K), generator=g, device='cuda').to(BF)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, 1)
  st = torch.cuda.current_stream().cuda_stream
  y, _ = ops.conv_fprop(d, x, w)
  yn = torch.empty_like(y)
  assert L.asm_conv2d_fprop_naive(C.byref(d), x.data_ptr(), w.data_ptr(), yn.data_ptr(), st) == 0
  assert util.rel_l2(y.float(), yn.float()) <= 4e-3
  wt = torch.empty((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, k, k, Cn)
  dx = ops.conv_dgrad(d, dy, wt)
  dxn = torch.empty_like(dx)
  assert L.asm_conv2d_dgrad_naive(C.byref(d), dy.data_ptr(), w.data_ptr(), dxn.data_ptr(), st) == 0
  assert util.rel_l2(dx.float(), dxn.float()) <= 4e-3
  dw = torch.empty((K, k, k, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, x, dy, dw)
  dwn = torch.empty_like(dw)
  assert L.asm_conv2d_wgrad_naive(C.byref(d), x.data_ptr(), dy.data_ptr(), dwn.data_ptr(), st) == 0
  assert util.rel_l2(dw, dwn) <= 2e-3
  # linearity (size-independent property): conv(x, 2w) == 2 conv(x, w) bit-exactly in bf16
  y2, _ = ops.conv_fprop(d, x, (w.float() * 2).to(BF))
  assert torch.equal(y2.float(), y.float() * 2)
  # wgrad is deterministic (slab reduce, no atomics)
  dw2 = torch.empty_like(dw)
  ops.conv_wgrad(d, x, dy, dw2)
  assert torch.equal(dw, dw2)


def test_tr_read_probe(hip_lib):
  """ds_read_b64_tr_b16 semantics the wgrad kernel relies on: with lane l reading LDS elements
  [4l, 4l+4) of lds[i] = i, lane l receives element j = 16*j + (l & 15) + 64*(l >> 4)."""
  from assembled_cnn_amd import ops
  out = torch.empty((64, 4), dtype=torch.int16, device='cuda')
  assert ops.L().asm_debug_tr_probe(out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
  lanes = torch.arange(64)[:, None]
  js = torch.arange(4)[None, :]
  expect = (16 * js + (lanes & 15) + 64 * (lanes >> 4)).to(torch.int16)
  assert torch.equal(out.cpu(), expect), out.cpu()


def test_error_convention(hip_lib):
  """ValueError / NotImplementedError split of the reference -> ASM_EINVAL / ASM_ENOTSUP."""
  from assembled_cnn_amd import ops
  x = torch.zeros((1, 4, 4, 12), dtype=BF, device='cuda')
  w = torch.zeros((8, 3, 3, 12), dtype=BF, device='cuda')
  with pytest.raises(ValueError):
    ops.conv_fprop(ops.make_conv_desc(1, 4, 4, 12, 8, 3, 3, 1), x, w)   # C % 8 != 0
  with pytest.raises(ValueError):
    ops.conv_fprop(ops.make_conv_desc(1, 4, 4, 16, 8, 3, 3, 3), x, w)   # stride 3


def test_batched_filter_transpose(hip_lib):
  """all CRSK copies of an arena in one launch == per-layer permutes (incl. a padded-K dense kernel)."""
  from assembled_cnn_amd import nn
  dev = torch.device('cuda')
  arena = nn.ParamArena()
  c = nn.Ctx(arena, True, True, 0.997, dev, False)
  convs = [nn.ConvKernel(c, 3, 64, 128), nn.ConvKernel(c, 1, 256, 64), nn.ConvKernel(c, 1, 2048, 1001, dense=True),
           nn.ConvKernel(c, 3, 32, 32)]
  arena.finalize(dev, 0)
  for cv in convs:
    w = arena.wb(cv.name)
    wt = arena.wt_view(cv._wts).view(cv.cin, cv.k, cv.k, cv.kpad)
    assert torch.equal(wt[..., :cv.cout], w.permute(3, 1, 2, 0))
    assert float(wt[..., cv.cout:].float().abs().sum()) == 0.0
  # the one-thread-per-element entry point gives the same arena
  from assembled_cnn_amd import ops
  tiled = arena.wt16.clone()
  arena.wt16.zero_()
  ops.filter_transpose_batched(arena.w16, arena.wt16, arena._wt_table, len(arena.wt_specs), arena._wt_total)
  assert torch.equal(arena.wt16, tiled)


@pytest.mark.parametrize('mode', [0, 1], ids=['igemm2', 'general-kernel'])
@pytest.mark.parametrize('tile', [3])
@pytest.mark.parametrize('shape', [(3, 7, 7, 256, 512, 3, 1), (2, 16, 16, 64, 128, 3, 1), (4, 20, 20, 128, 320, 1, 1),
                                   (2, 14, 14, 128, 256, 3, 2)], ids=lambda s: 'x'.join(map(str, s)))
def test_big_tile_variants(hip_lib, shape, tile, mode, monkeypatch):
  """the 256x256 (8-wave) tile configuration, forced through ASM_IGEMM_TILE, incl. ragged M, masked N
  and the two-partials-per-tile statistics epilogue."""
  from assembled_cnn_amd import ops
  util.set_knob(monkeypatch, 'ASM_IGEMM_TILE', str(tile))
  if mode:
    util.set_knob(monkeypatch, 'ASM_IGEMM_MODE', str(mode))
  N, H, W, Cn, K, k, stride = shape
  x = _rand((N, H, W, Cn), 21)
  w = _rand((K, k, k, Cn), 22, scale=(1.0 / (k * k * Cn)) ** 0.5)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  y, stats = ops.conv_fprop(d, x.cuda(), w.cuda(), want_stats=True)
  ref = _ref_conv(x, w, stride)
  _check(y, ref, name='fprop tile %d' % tile)
  yb = y.float().view(-1, K)
  M = yb.shape[0]
  assert stats.shape[0] == (M + 127) // 128
  for b in range(stats.shape[0]):
    blk = yb[b * 128:(b + 1) * 128]
    assert torch.allclose(stats[b, 0], blk.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats[b, 1], (blk * blk).sum(0), rtol=1e-4, atol=1e-2)
  dy = _rand(tuple(ref.shape), 23)
  wt = torch.zeros((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(w.cuda(), wt, K, k, k, Cn)
  dx = ops.conv_dgrad(d, dy.cuda(), wt)
  util.set_knob(monkeypatch, 'ASM_IGEMM_TILE', '1')
  dx1 = ops.conv_dgrad(d, dy.cuda(), wt)
  y1, _ = ops.conv_fprop(d, x.cuda(), w.cuda())
  # same K-order of accumulation in every tile config -> identical bits
  assert torch.equal(dx, dx1) and torch.equal(y, y1)


@pytest.mark.parametrize('shape', [(2, 14, 14, 64, 128, 3, 1), (3, 7, 7, 256, 512, 1, 1), (2, 16, 16, 32, 64, 3, 2),
                                   (2, 14, 14, 256, 512, 3, 1)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('res,relu', [(False, True), (True, True), (True, False)])
def test_fprop_with_folded_inference_bn(hip_lib, shape, res, relu):
  """asm_conv2d_fprop_bn == conv -> bn_apply(moving statistics) [-> + residual] [-> relu], up to the one rounding
  it saves, and == the fp32 reference."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, k, stride = shape
  x = _rand((N, H, W, Cn), 31)
  w = _rand((K, k, k, Cn), 32, scale=(1.0 / (k * k * Cn)) ** 0.5)
  g = torch.Generator().manual_seed(33)
  scale = (torch.rand(K, generator=g) + 0.5)
  shift = torch.randn(K, generator=g) * 0.2
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  ref = _ref_conv(x, w, stride)                                  # fp32 NHWC
  r = _rand(tuple(ref.shape), 34) if res else None
  want = ref * scale + shift + (r.float() if res else 0.0)
  if relu:
    want = want.clamp(min=0)
  y = ops.conv_fprop_bn(d, x.cuda(), w.cuda(), scale.cuda(), shift.cuda(), r.cuda() if res else None, relu)
  _check(y, want, name='fprop_bn vs fp32')
  y0, _ = ops.conv_fprop(d, x.cuda(), w.cuda())
  M = y0.numel() // K
  two = ops.bn_apply(y0.view(M, K), M, K, scale.cuda(), shift.cuda(), r.cuda().view(M, K) if res else None,
                     1 if res else 0, relu, d.Ho, d.Wo)
  assert util.rel_l2(y.float().cpu().view(M, K), two.float().cpu()) <= 6e-3
  with pytest.raises(ValueError):
    ops.conv_fprop_bn(ops.make_conv_desc(N, H, W, Cn, K, k, k, stride, out_f32=True), x.cuda(), w.cuda(), scale.cuda(),
                      shift.cuda())


@pytest.mark.parametrize('shape', [(4, 14, 14, 128, 512, 1, 1), (2, 28, 28, 64, 256, 1, 1), (2, 14, 14, 64, 64, 3, 1),
                                   (32, 14, 14, 256, 1024, 1, 1)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('pfa', [0, 1], ids=['addend-inline', 'addend-prefetched'])
def test_dgrad_with_masked_addend(hip_lib, shape, pfa, monkeypatch):
  """asm_conv2d_dgrad_masked(addend, mask) against the oracle's input gradient + addend * [mask bit]; == asm_conv2d_dgrad(
  addend * mask) bit for bit, and asm_mask_apply is that product: the lazily masked shortcut gradient is the same gradient.  Both epilogue variants (addend fetched inside the
  store passes / prefetched ahead of them, ASM_IGEMM_PFA) and in-place accumulation (dx == addend)."""
  from assembled_cnn_amd import ops
  util.set_knob(monkeypatch, 'ASM_IGEMM_PFA', str(pfa))
  N, H, W, Cn, K, k, stride = shape
  g = torch.Generator(device='cuda').manual_seed(17)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  dy = torch.randn((N, d.Ho, d.Wo, K), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, k, k, Cn), generator=g, device='cuda') * (k * k * Cn) ** -0.5).to(BF)
  wt = torch.empty((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, k, k, Cn)
  addend = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF)
  mask = torch.randint(0, 256, (N * H * W, Cn // 8), generator=g, device='cuda', dtype=torch.uint8)
  masked = ops.mask_apply(addend, mask)
  bits = ((mask.to(torch.int32)[:, :, None] >> torch.arange(8, device='cuda', dtype=torch.int32)) & 1).reshape(N, H, W, Cn)
  assert torch.equal(masked.float(), addend.float() * bits.float())
  a = ops.conv_dgrad(d, dy, wt, addend, mask)
  b = ops.conv_dgrad(d, dy, wt, masked)
  assert torch.equal(a, b)
  # against the oracle: the convolution's input gradient (autograd of its conv2d_fixed_padding) + addend * [mask bit]
  from oracle import assembled_oracle as O
  xr = torch.zeros((N, H, W, Cn), requires_grad=True)
  yr = O._conv_raw(xr.permute(0, 3, 1, 2), w.float().cpu().permute(1, 2, 3, 0), k, stride)
  gx, = torch.autograd.grad(yr, [xr], dy.float().cpu().permute(0, 3, 1, 2))
  _check(a, gx + addend.float().cpu() * bits.float().cpu(), name='dgrad + masked addend vs oracle')
  util.set_knob(monkeypatch, 'ASM_IGEMM_PFA', str(1 - pfa))
  assert torch.equal(a, ops.conv_dgrad(d, dy, wt, addend, mask)), 'the two epilogue variants must agree bit for bit'
  # in-place fan-in accumulation through the C ABI: dx aliases the addend
  from assembled_cnn_amd.ops import L, _ptr, _stream, check
  import ctypes as C
  inplace = masked.clone()
  check(L().asm_conv2d_dgrad(C.byref(d), _ptr(dy), _ptr(wt), _ptr(inplace), _ptr(inplace), _stream()), 'dgrad in place')
  assert torch.equal(inplace, b)


@pytest.mark.parametrize('shape,pool', [
    ((4, 28, 28, 256, 64, 1, 1), (3, 2, 1, False)),     # BigLittle projection: 3x3 / 2, pad 1, divisor 9
    ((2, 14, 14, 512, 128, 1, 1), (2, 2, 0, False)),    # ResNet-D: 2x2 / 2
    ((2, 14, 14, 64, 32, 1, 1), (2, 1, 0, True)),       # ResNet-D stride 1: SAME 2x2, valid-count divisor
    ((32, 14, 14, 1024, 256, 1, 1), (3, 2, 1, False)),  # 256 x 256 tile path
], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('masked', [False, True], ids=['plain', 'masked-addend'])
def test_dgrad_with_pooled_gradient_gathered_in_the_epilogue(hip_lib, shape, pool, masked):
  """asm_conv2d_dgrad_pooled against the oracle (fp32: the 1x1 input gradient + addend * [mask bit] + the autograd gradient of
  the oracle's average pool), and == asm_conv2d_dgrad[_masked] followed by asm_avgpool_bwd(addend = that) up to one bf16
  rounding (the fused form rounds the sum once)."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, k, stride = shape
  pk, pst, ppad, cv = pool
  g = torch.Generator(device='cuda').manual_seed(23)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  dy = torch.randn((N, H, W, K), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, 1, 1, Cn), generator=g, device='cuda') * Cn ** -0.5).to(BF)
  wt = torch.empty((Cn, 1, 1, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, 1, 1, Cn)
  if cv:
    Hp, Wp = H, W
  else:
    Hp, Wp = (H + (pk - 1) - pk) // pst + 1, (W + (pk - 1) - pk) // pst + 1
  dpool = torch.randn((N, Hp, Wp, Cn), generator=g, device='cuda').to(BF)
  addend = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF) if masked else None
  mask = torch.randint(0, 256, (N * H * W, Cn // 8), generator=g, device='cuda', dtype=torch.uint8) if masked else None
  assert ops.dgrad_pool_ok(d)
  fused = ops.conv_dgrad(d, dy, wt, addend, mask, pool=(dpool, pk, pst, ppad, cv))
  two = ops.conv_dgrad(d, dy, wt, addend, mask)
  two = ops.avgpool_bwd(dpool, (N, H, W, Cn), pk, pst, ppad, cv, addend=two)
  r = util.rel_l2(fused.float().cpu(), two.float().cpu())
  assert r <= 3e-3, r
  # fp32 reference of the sum
  ref = dy.float().cpu().view(-1, K) @ w.float().cpu().view(K, Cn)
  if masked:
    bits = ((mask.cpu().to(torch.int32)[:, :, None] >> torch.arange(8, dtype=torch.int32)) & 1).reshape(-1, Cn)
    ref = ref + addend.float().cpu().view(-1, Cn) * bits
  # the pooled part from the ORACLE's average pool (autograd), not from the product's scatter kernel
  from oracle import assembled_oracle as O
  xr = torch.zeros((N, Cn, H, W), requires_grad=True)
  pr = O.avg_pool_same(xr, pk, pst) if cv else O.avg_pool_valid(O.fixed_padding(xr, pk) if ppad else xr, pk, pst)
  assert tuple(pr.shape) == (N, Cn, Hp, Wp)
  pz, = torch.autograd.grad(pr, [xr], dpool.float().cpu().permute(0, 3, 1, 2))
  ref = ref + pz.permute(0, 2, 3, 1).reshape(-1, Cn)
  assert util.rel_l2(fused.float().cpu().view(-1, Cn), ref) <= 4e-3
  assert util.max_abs(fused.float().cpu().view(-1, Cn), ref) <= float(ref.abs().max()) * 2 ** -7 + 1e-6


IGEMM3_SHAPES = [(2, 14, 14, 128, 256), (5, 7, 7, 192, 136), (1, 9, 30, 128, 128), (3, 5, 11, 256, 72), (4, 28, 28, 128, 256),
                 (16, 14, 14, 512, 1024),
                 # single 64-channel chunk (round 5: one row buffer; 192 rows up to W = 30, 256 rows up to W = 62)
                 (2, 56, 56, 64, 128), (3, 28, 28, 64, 128), (1, 11, 62, 64, 72), (2, 9, 31, 64, 136)]


@pytest.mark.parametrize('shape', IGEMM3_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_igemm3_is_igemm2_bit_for_bit(hip_lib, shape, monkeypatch):
  """igemm3_kernel keeps the activation rows resident across the nine taps instead of re-staging a tile per tap; the
  products are accumulated in the same (chunk, tap, k) order as igemm2_kernel's, so every output (forward with its fused
  statistics, input gradient plain / with a fan-in addend / with a masked addend) must be IDENTICAL, border pixels, ragged
  row tiles and channel tails included."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K = shape
  x = _rand((N, H, W, Cn), 11).cuda()
  w = _rand((K, 3, 3, Cn), 12, scale=(1.0 / (9 * Cn)) ** 0.5).cuda()
  dy = _rand((N, H, W, K), 13).cuda()
  addend = _rand((N, H, W, Cn), 14).cuda()
  mask = torch.randint(0, 256, (N, H, W, Cn // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(15)).cuda()
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 1)
  wt = torch.zeros((Cn, 3, 3, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, 3, 3, Cn)
  outs = {}
  for knob in ('2' if Cn != 64 else '3', '0'):
    util.set_knob(monkeypatch, 'ASM_IGEMM3', knob)
    y, st = ops.conv_fprop(d, x, w, want_stats=True)
    y2, _ = ops.conv_fprop(d, x, w, want_stats=False)
    dx = ops.conv_dgrad(d, dy, wt)
    dxa = ops.conv_dgrad(d, dy, wt, addend=addend)
    dxm = ops.conv_dgrad(d, dy, wt, addend=addend, addend_mask=mask)
    outs['2' if knob != '0' else '0'] = (y, st, y2, dx, dxa, dxm)
  for a, b, name in zip(outs['2'], outs['0'], ('fprop', 'stats', 'fprop-nostats', 'dgrad', 'dgrad+addend', 'dgrad+masked')):
    assert torch.equal(a, b), name


@pytest.mark.parametrize('shape', [(2, 16, 16), (3, 32, 48), (1, 112, 112)], ids=lambda s: 'x'.join(map(str, s)))
def test_one_launch_stride2_dgrad_is_the_parity_class_launches_bit_for_bit(hip_lib, shape, monkeypatch):
  """dgrad_s2_kernel (3x3 / stride 2, 64 -> 64: filter slice in registers, the four parity classes of dx side by side,
  fan-in addend / masked addend in the copy-out) accumulates in the order of the four parity-class launches it replaces,
  so every output must be IDENTICAL to theirs; and both must match the oracle's autograd."""
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  N, H, W = shape
  Cn = K = 64
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 2)
  w = _rand((K, 3, 3, Cn), 21, scale=(1.0 / (9 * Cn)) ** 0.5)
  dy = _rand((N, d.Ho, d.Wo, K), 22)
  addend = _rand((N, H, W, Cn), 23).cuda()
  mask = torch.randint(0, 256, (N, H, W, Cn // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(24)).cuda()
  wt = torch.zeros((Cn, 3, 3, K), dtype=BF, device='cuda')
  ops.filter_transpose(w.cuda(), wt, K, 3, 3, Cn)
  outs = {}
  for knob in ('1', '0'):
    util.set_knob(monkeypatch, 'ASM_DGRAD_PARITY', '2' if knob == '1' else '1')    # one launch / four parity-class launches
    assert ops.dgrad_s2_ok(d) == (knob == '1')
    outs[knob] = (ops.conv_dgrad(d, dy.cuda(), wt), ops.conv_dgrad(d, dy.cuda(), wt, addend=addend),
                  ops.conv_dgrad(d, dy.cuda(), wt, addend=addend, addend_mask=mask))
  for a, b, name in zip(outs['1'], outs['0'], ('plain', 'addend', 'masked addend')):
    assert torch.equal(a, b), name
  xr = torch.zeros((N, Cn, H, W), requires_grad=True)
  yr = O._conv_raw(xr, w.float().permute(1, 2, 3, 0), 3, 2)
  (gx,) = torch.autograd.grad(yr, [xr], dy.float().permute(0, 3, 1, 2))
  _check(outs['1'][0], gx.permute(0, 2, 3, 1), name='stride-2 dgrad vs oracle')


@pytest.mark.parametrize('shape,splits', [((4, 14, 14, 128, 256), 0), ((2, 14, 14, 512, 1024), 0), ((8, 7, 7, 256, 512), 0),
                                          ((3, 7, 7, 64, 128), 0), ((5, 14, 14, 64, 128), 1), ((6, 7, 14, 128, 128), 3)],
                         ids=lambda s: 'x'.join(map(str, s)) if isinstance(s, tuple) else 'splits%d' % s)
def test_deep_3x3_weight_gradient_vs_fp64_and_oracle(hip_lib, shape, splits, monkeypatch):
  """The weight gradient of the deep 3x3 stride-1 layers (14- and 7-wide maps) against fp64 sums of the bf16 products and
  against the ORACLE (autograd of its conv2d_fixed_padding with respect to the filter).  Border pixels are the point: images
  end inside 64-pixel steps (196 and 49 pixels per image), pixel ranges end inside images, a pixel range per split, one split
  (straight into dW), a non-square map."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K = shape
  x = _rand((N, H, W, Cn), 31).cuda()
  dy = _rand((N, H, W, K), 32).cuda()
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 1)
  if splits:
    util.set_knob(monkeypatch, 'ASM_WGRAD_SPLITS', str(splits))
  dw = torch.full((K, 3, 3, Cn), float('nan'), device='cuda')
  ops.conv_wgrad(d, x, dy, dw)
  torch.cuda.synchronize()
  # fp64 reference: dW[k][r][s][c] = sum_{n,h,w} dy[n,h,w,k] * x[n,h+r-1,w+s-1,c] with zero padding
  xp = torch.nn.functional.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1))
  ref = torch.empty((K, 3, 3, Cn), dtype=torch.float64, device='cuda')
  dyd = dy.double().reshape(-1, K)
  for r in range(3):
    for s_ in range(3):
      xs = xp[:, :, r:r + H, s_:s_ + W].permute(0, 2, 3, 1).reshape(-1, Cn)
      ref[:, r, s_, :] = dyd.t() @ xs
  assert bool(torch.isfinite(dw).all())
  assert float((dw.double() - ref).abs().max()) / float(ref.abs().max()) <= 2e-5
  from oracle import assembled_oracle as O
  wr = torch.zeros((3, 3, Cn, K), requires_grad=True)
  yr = O._conv_raw(x.float().cpu().permute(0, 3, 1, 2), wr, 3, 1)
  gw, = torch.autograd.grad(yr, [wr], dy.float().cpu().permute(0, 3, 1, 2))
  assert util.rel_l2(dw.cpu().permute(1, 2, 3, 0), gw) <= 1e-4


GEMM1_CODES = (1, 5, 8, 10, 11, 12, 13, 14, 16)


@pytest.mark.parametrize('shape', [(2, 14, 14, 256, 512, 1), (3, 7, 7, 1024, 264, 1), (2, 28, 28, 64, 72, 1), (1, 9, 11, 40, 136, 1),
                                   (2, 28, 28, 128, 256, 2)], ids=lambda s: 'x'.join(map(str, s)))
def test_ring_gemm_of_the_1x1_layers_is_igemm2_bit_for_bit(hip_lib, shape, monkeypatch):
  """igemm1_kernel (csrc/conv_gemm1.hip: the 1x1 layers as a GEMM with a ring of LDS stages and counted vmcnt) multiplies in
  igemm2_kernel's (chunk, k) order with its epilogue: every tile / depth of its table must give IDENTICAL outputs --
  forward, input gradient plain / with a fan-in addend / with a masked addend / with the pooled-gradient gather -- ragged
  row tiles, channel tails (a single-chunk reduction of 40 channels), output tails and the stride-2 projection (parity
  class, strided output) included; the fused statistics (summed per tile width) within fp32 summation order; and the
  forward output against the oracle's convolution."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, st = shape
  d = ops.make_conv_desc(N, H, W, Cn, K, 1, 1, st)
  x = _rand((N, H, W, Cn), 31).cuda()
  w = _rand((K, 1, 1, Cn), 32, scale=Cn ** -0.5).cuda()
  dy = _rand((N, d.Ho, d.Wo, K), 33).cuda()
  addend = _rand((N, H, W, Cn), 34).cuda()
  mask = torch.randint(0, 256, (N, H, W, Cn // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(35)).cuda()
  wt = torch.zeros((Cn, 1, 1, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, 1, 1, Cn)
  pooled = st == 1 and K % 32 == 0 and H % 2 == 0 and W % 2 == 0
  pdy = _rand((N, H // 2, W // 2, Cn), 36).cuda() if pooled else None

  def run():
    y, stt = ops.conv_fprop(d, x, w, want_stats=True)
    outs = [y, stt, ops.conv_dgrad(d, dy, wt), ops.conv_dgrad(d, dy, wt, addend=addend)]
    if st == 1:
      outs.append(ops.conv_dgrad(d, dy, wt, addend=addend, addend_mask=mask))
    if pooled:
      outs.append(ops.conv_dgrad(d, dy, wt, addend=addend, addend_mask=mask, pool=(pdy, 2, 2, 0, False)))
    return outs
  util.set_knob(monkeypatch, 'ASM_GEMM1', '0')
  ref = run()
  yo = _ref_conv(x.cpu(), w.cpu(), st)
  for code in GEMM1_CODES:
    util.set_knob(monkeypatch, 'ASM_GEMM1', str(code))
    poison = [torch.full_like(t, float('nan')) for t in ref for _ in range(2)]
    del poison        # the caching allocator hands these blocks to run(): an output that is not written cannot pass on stale bytes
    got = run()
    for i, (a, b) in enumerate(zip(got, ref)):
      if i == 1:
        assert float((a.double() - b.double()).norm()) <= 1e-5 * float(b.double().norm()) + 1e-6, ('stats', code)
      else:
        assert torch.equal(a, b), (i, code)
  util.set_knob(monkeypatch, 'ASM_GEMM1', '-1')
  _check(ref[0], yo, name='1x1 forward vs oracle')


@pytest.mark.parametrize('shape', [(2, 14, 14, 256, 512), (3, 7, 7, 520, 264), (1, 28, 28, 64, 48), (2, 9, 11, 40, 24)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_ring_weight_gradient_is_the_register_staged_one_bit_for_bit(hip_lib, shape, monkeypatch):
  """wgrad_kernel<.., LIN, 2>: the two tiles of a 64-pixel step by LDS-DMA into the other of two stages (source-side swizzle
  of the transposing reads) instead of through registers -- the same steps in the same order under the same pixel split,
  so dW must be IDENTICAL, pixel tails and channel tails included; and within fp32 noise of a float64
  product."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K = shape
  d = ops.make_conv_desc(N, H, W, Cn, K, 1, 1, 1)
  x = _rand((N, H, W, Cn), 41).cuda()
  dy = _rand((N, H, W, K), 42).cuda()
  outs = {}
  for splits in ('0', '3'):
    util.set_knob(monkeypatch, 'ASM_WGRAD_SPLITS', splits)
    for ring in ('0', '2'):
      util.set_knob(monkeypatch, 'ASM_WGRAD_RING', ring)
      dw = torch.full((K, 1, 1, Cn), float('nan'), dtype=torch.float32, device='cuda')
      ops.conv_wgrad(d, x, dy, dw)
      outs[(splits, ring)] = dw
    assert torch.equal(outs[(splits, '2')], outs[(splits, '0')]), splits
  want = dy.double().reshape(-1, K).t() @ x.double().reshape(-1, Cn)
  got = outs[('0', '2')].double().reshape(K, Cn)
  assert float((got - want).norm() / want.norm()) <= 1e-5
  # and against the oracle: autograd of its convolution with respect to the filter
  from oracle import assembled_oracle as O
  wr = torch.zeros((1, 1, Cn, K), requires_grad=True)
  yr = O._conv_raw(x.float().cpu().permute(0, 3, 1, 2), wr, 1, 1)
  gw, = torch.autograd.grad(yr, [wr], dy.float().cpu().permute(0, 3, 1, 2))
  assert util.rel_l2(outs[('0', '2')].cpu().permute(1, 2, 3, 0), gw) <= 1e-4


IGEMM8_SHAPES = [
    # N, H,  W,  C,   K     (3x3, stride 1)
    (2, 14, 14, 128, 256),    # M = 392: one full + one ragged 256-row tile; 2 chunks (18 steps)
    (3, 7, 7, 192, 448),      # M = 147 (< one tile), 3 / 7 chunks (odd: forward / input gradient), ragged second N tile (448, 192)
    (1, 16, 16, 64, 512),     # ONE chunk (nine steps), exactly one row tile, two N tiles
    (5, 9, 13, 256, 320),     # odd H / W, images inside a tile, 4 chunks, N tail 64 of 256
    (24, 14, 14, 512, 256),   # 8 chunks (72 steps), 19 row tiles
]


@pytest.mark.parametrize('shape', IGEMM8_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_igemm8_bit_identical_to_igemm2_and_vs_oracle(hip_lib, shape, monkeypatch):
  """igemm8_kernel (wave-staggered multi-phase main loop, csrc/conv_igemm8.hip) against the oracle AND bit for bit against
  igemm2_kernel's 256 x 256 tile (same operands, same (chunk, tap, k) accumulation order): forward with / without the fused
  statistics, input gradient with / without the fan-in addend.  Repeated launches must agree with each other: a staging race
  (a half-tile read before its DMA landed, or re-staged before its last read) shows as run-to-run differences."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K = shape
  x = _rand((N, H, W, Cn), 11)
  w = _rand((K, 3, 3, Cn), 12, scale=(1.0 / (9 * Cn)) ** 0.5)
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 1)
  xd, wd = x.cuda(), w.cuda()
  ref = _ref_conv(x, w, 1)
  dy = _rand(tuple(ref.shape), 13).cuda()
  wt = torch.zeros((Cn, 3, 3, K), dtype=BF, device='cuda')
  ops.filter_transpose(wd, wt, K, 3, 3, Cn)
  add = _rand((N, H, W, Cn), 14).cuda()

  def run():
    y, st = ops.conv_fprop(d, xd, wd, want_stats=True)
    y2, _ = ops.conv_fprop(d, xd, wd, want_stats=False)
    dx = ops.conv_dgrad(d, dy, wt)
    dxa = ops.conv_dgrad(d, dy, wt, addend=add)
    return y, st, y2, dx, dxa

  util.set_knob(monkeypatch, 'ASM_IGEMM8', '0')
  util.set_knob(monkeypatch, 'ASM_IGEMM_TILE', '3')      # igemm2_kernel<256, 256, 64>
  base = run()
  assert hip_lib.asm_debug_last_conv_kernel() == 2
  util.set_knob(monkeypatch, 'ASM_IGEMM_TILE', '0')
  util.set_knob(monkeypatch, 'ASM_IGEMM8', '2')          # igemm8_kernel wherever the shape allows
  for rep in range(6):
    got = run()
    assert hip_lib.asm_debug_last_conv_kernel() == 8, 'the layer did not run on igemm8_kernel'
    for name, a, b in zip(('fprop+stats', 'stats', 'fprop', 'dgrad', 'dgrad+addend'), got, base):
      assert torch.equal(a, b), 'igemm8 %s differs from igemm2 (rep %d): max |d| %.3e' % (
          name, rep, float((a.float() - b.float()).abs().max()))
  _check(got[0], ref, name='igemm8 fprop')
  xr = x.float().requires_grad_(True)
  from oracle import assembled_oracle as O
  yr = O._conv_raw(xr.permute(0, 3, 1, 2), w.float().permute(1, 2, 3, 0), 3, 1)
  gx, = torch.autograd.grad(yr, [xr], dy.float().cpu().permute(0, 3, 1, 2))
  _check(got[3], gx, name='igemm8 dgrad')


def test_igemm8_ragged_last_round_split(hip_lib, monkeypatch):
  """260 tiles of 256 x 256 on a 256-CU chip: the 256 tiles of the full round run on igemm8_kernel, the rows of the last
  four on the 128-row kernel (conv_igemm.hip, "the ragged last round").  Forward (+ statistics) and input gradient
  (+ addend) must equal the unsplit igemm2 result bit for bit, statistics partial rows included."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K = 65, 16, 16, 128, 1024
  x = _rand((N, H, W, Cn), 21)
  w = _rand((K, 3, 3, Cn), 22, scale=(1.0 / (9 * Cn)) ** 0.5)
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 1)
  xd, wd = x.cuda(), w.cuda()
  util.set_knob(monkeypatch, 'ASM_IGEMM8', '0')
  y0, st0 = ops.conv_fprop(d, xd, wd, want_stats=True)
  assert hip_lib.asm_debug_last_conv_kernel() == 2
  util.set_knob(monkeypatch, 'ASM_IGEMM8', '1')
  n0 = hip_lib.asm_launch_count()
  y1, st1 = ops.conv_fprop(d, xd, wd, want_stats=True)
  assert hip_lib.asm_launch_count() - n0 == 2 and hip_lib.asm_debug_last_conv_kernel() == 3, 'expected igemm8 + igemm3'
  assert torch.equal(y0, y1) and torch.equal(st0, st1)
  # the same rows with the wide operand on the other side: K = 128 -> C = 1024 input gradient of a 1024 -> 128 layer
  d2 = ops.make_conv_desc(N, H, W, K, Cn, 3, 3, 1)
  dy = _rand((N, H, W, Cn), 23).cuda()
  w2 = _rand((Cn, 3, 3, K), 24, scale=(1.0 / (9 * K)) ** 0.5).cuda()
  wt = torch.zeros((K, 3, 3, Cn), dtype=BF, device='cuda')
  ops.filter_transpose(w2, wt, Cn, 3, 3, K)
  add = _rand((N, H, W, K), 25).cuda()
  util.set_knob(monkeypatch, 'ASM_IGEMM8', '0')
  g0 = ops.conv_dgrad(d2, dy, wt, addend=add)
  util.set_knob(monkeypatch, 'ASM_IGEMM8', '1')
  n0 = hip_lib.asm_launch_count()
  g1 = ops.conv_dgrad(d2, dy, wt, addend=add)
  assert hip_lib.asm_launch_count() - n0 == 2
  assert torch.equal(g0, g1)


BNRED_CASES = [
    # N, H,  W,  C(dx), K(dy), k, knobs                       what runs
    (4, 14, 14, 256, 64, 1, {}),                              # igemm2's statistics epilogue (not a workload shape), 8 row tiles
    (256, 14, 14, 512, 128, 1, {}),                           # a workload shape: igemm1 (configuration from the table)
    (3, 9, 11, 264, 72, 1, {}),                               # ragged rows, a channel tail in the third N tile
    (2, 9, 11, 72, 40, 1, {'ASM_IGEMM_MODE': '1'}),           # the general kernel, channel tails
    (2, 14, 14, 128, 256, 3, {}),                             # 3x3 on igemm3_kernel: 4 chunks, ragged last row tile
    (3, 7, 7, 192, 128, 3, {}),                               # igemm3, two chunks, M = 147 (two row tiles, the second ragged)
    (2, 28, 28, 128, 64, 3, {}),                              # igemm3's single-chunk form (one halo buffer)
    (24, 14, 14, 512, 256, 3, {'ASM_IGEMM8': '2'}),           # igemm8_kernel (forced): 19 row tiles, two column tiles
    (3, 7, 7, 448, 192, 3, {'ASM_IGEMM8': '2'}),              # igemm8, M = 147 < one tile, N tail 192 of 256 ... 3 chunks
]


@pytest.mark.parametrize('case', BNRED_CASES, ids=lambda c: 'x'.join(map(str, c[:6])))
@pytest.mark.parametrize('relu', [True, False])
def test_dgrad_with_bn_backward_sums_in_its_epilogue(hip_lib, case, relu, monkeypatch):
  """asm_conv2d_dgrad_bnred: the input gradient that also reduces (sum dz, sum dz * y) of the batch norm behind its output (1x1 layers, and 3x3 layers on igemm3 / igemm8).
  dx must be the bits asm_conv2d_dgrad[_masked] writes; the partial rows must sum to the sums of the bf16 dx it wrote; and
  the batch-norm backward finished from them (asm_bn_bwd_finalize_raw + apply) must agree with the three-pass form
  (reduce over (dx, y) + finalize + apply): dgamma / dbeta to 1e-3, dy rel-L2 <= 2e-3."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, k, knobs = case
  for name, val in knobs.items():
    util.set_knob(monkeypatch, name, val)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, 1)
  g = torch.Generator(device='cuda').manual_seed(5)
  dy = torch.randn((N, H, W, K), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, k, k, Cn), generator=g, device='cuda') * (k * k * K) ** -0.5).to(BF)
  wt = torch.zeros((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, k, k, Cn)
  add = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF)
  amask = torch.randint(0, 256, (N, H, W, Cn // 8), generator=g, device='cuda', dtype=torch.uint8)
  y = (torch.randn((N, H, W, Cn), generator=g, device='cuda') * 1.5 + 0.7).to(BF)     # a mean well away from 0
  rmask = torch.randint(0, 256, (N * H * W, Cn // 8), generator=g, device='cuda', dtype=torch.uint8) if relu else None
  M = N * H * W
  for addend, mask in ((None, None), (add, None), (add, amask)):
    ref = ops.conv_dgrad(d, dy, wt, addend, mask)
    kern = hip_lib.asm_debug_last_conv_kernel()
    dx, part = ops.conv_dgrad_bnred(d, dy, wt, addend, mask, y, rmask)
    assert hip_lib.asm_debug_last_conv_kernel() == kern, 'the fused form left the kernel family'
    assert torch.equal(dx, ref), 'dx differs from the plain input gradient'
    dz = dx.float().view(M, Cn)
    if relu:
      bits = ((rmask.to(torch.int32)[..., None] >> torch.arange(8, dtype=torch.int32, device='cuda')) & 1).view(M, Cn)
      dz = dz * bits
    s0, s1 = dz.double().sum(0), (dz.double() * y.double().view(M, Cn)).sum(0)
    got = part.double().sum(0)
    assert part.shape[0] == (M + 127) // 128
    scale0, scale1 = float(dz.abs().double().sum(0).max()), float((dz.abs().double() * y.double().view(M, Cn).abs()).sum(0).max())
    assert float((got[0] - s0).abs().max()) <= 1e-5 * scale0 and float((got[1] - s1).abs().max()) <= 1e-5 * scale1
  # the batch-norm backward from these sums vs its own reduce pass
  mean, var = y.float().view(M, Cn).mean(0), y.float().view(M, Cn).var(0, unbiased=False)
  invstd = (var + 1e-5).rsqrt()
  gamma = (torch.rand(Cn, generator=g, device='cuda') + 0.5)
  dg0, db0, dg1, db1 = (torch.empty(Cn, device='cuda') for _ in range(4))
  yv = y.view(M, Cn)
  a0, _ = ops.bn_bwd(dx.view(M, Cn), yv, rmask, relu, M, Cn, gamma, mean, invstd, dg0, db0, False)
  a1, _ = ops.bn_bwd(dx.view(M, Cn), yv, rmask, relu, M, Cn, gamma, mean, invstd, dg1, db1, False, raw_part=part)
  assert torch.allclose(dg0, dg1, rtol=1e-3, atol=1e-3 * float(dg0.abs().max()))
  assert torch.allclose(db0, db1, rtol=1e-3, atol=1e-3 * float(db0.abs().max()))
  assert util.rel_l2(a1.float().cpu(), a0.float().cpu()) <= 2e-3


def test_dgrad_bnred_refuses_what_it_does_not_cover(hip_lib):
  """3x3 layers outside igemm8 / igemm3 (64 or fewer output channels of the gradient; stride 2) have no kernel with the sums"""
  from assembled_cnn_amd import ops
  for d in (ops.make_conv_desc(2, 16, 32, 64, 32, 3, 3, 1), ops.make_conv_desc(2, 9, 9, 64, 64, 3, 3, 1),
            ops.make_conv_desc(2, 16, 16, 128, 128, 3, 3, 2),
            ops.make_conv_desc(2, 8, 8, 128, 72, 3, 3, 1)):
    assert hip_lib.asm_conv2d_dgrad_bnred_blocks(C.byref(d)) == 0
    assert hip_lib.asm_conv2d_dgrad_bnred(C.byref(d), 1, 1, None, None, 1, None, 1, 1, None) == -2    # ASM_ENOTSUP before any pointer is touched
    assert not ops.dgrad_bnred_ok(d)
  assert ops.dgrad_bnred_ok(ops.make_conv_desc(2, 14, 14, 128, 256, 3, 3, 1))
