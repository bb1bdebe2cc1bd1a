"""GPU parity AT THE BENCHMARK CONFIGURATION.

Every distinct convolution of the two BASELINE training configurations (Assemble-ResNet-50 + D and ResNet-50 v1.5,
batch 256, 224x224; the list comes from walking the product model in its shape-only mode, tools/list_convs.py) is run
through the C ABI with the kernel plan the library picks for that shape -- tile sizes, 256 x 256 / 8-wave kernels,
parity-class dgrad, pixel splits -- and compared

  * at N = 256 (the full tensor) with the one-thread-per-output direct convolutions of include/asm_hip_debug.h
    (plain loops over the definition of conv2d_fixed_padding, nets/model_helper.py:67-78), fprop / dgrad / wgrad;
  * at N = 8 with the CPU oracle (and its autograd), which also pins the direct kernels to the oracle.

Separately the 256 x 256 / 8-wave weight-gradient kernel -- reached only by >= 40 GFLOP layers -- is forced onto small
shapes (ragged pixel count, padded column tiles, forced pixel splits) and checked against the oracle, and checked
against the direct kernel on a real plan.

Tolerances: bf16 outputs of two fp32-accumulating kernels differ by isolated 1-ulp roundings: relative L2 <= 3e-3,
max |err| <= 2^-7 max|ref|; fp32 weight gradients: relative L2 <= 2e-3 (summation order only)."""
import ctypes as C
import os
import sys

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _shapes(workload, batch):
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  import list_convs
  # the dense layer (K = 1001, fp32 logits with a padded row stride) has its own test in tests/test_gpu_conv.py
  return [k for k in list_convs.conv_shapes(workload, batch).keys() if k[4] % 8 == 0]


def _close_bf16(a, b, what, errs, rel=3e-3):
  r = util.rel_l2(a.float(), b.float())
  m = float((a.float() - b.float()).abs().max())
  lim = float(b.float().abs().max()) * 2 ** -7 + 1e-6
  if not (r <= rel and m <= lim):
    errs.append('%s: rel_l2 %.3e (<= %.1e) max_abs %.3e (<= %.3e)' % (what, r, rel, m, lim))


def _close_f32(a, b, what, errs, rel=2e-3):
  r = util.rel_l2(a, b)
  if not r <= rel:
    errs.append('%s: rel_l2 %.3e (<= %.1e)' % (what, r, rel))


def _plan(L, d):
  arr = (C.c_int32 * 6)()
  assert L.asm_conv2d_wgrad_plan(C.byref(d), C.byref(arr)) == 0
  return list(arr)


def _run_generic(L, key, errs, against_oracle):
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, R, S, stride, _ = key
  tag = 'N%d %dx%d C%d->K%d %dx%d/%d' % (N, H, W, Cn, K, R, S, stride)
  g = torch.Generator(device='cuda').manual_seed(hash(key) & 0xffff)
  d = ops.make_conv_desc(N, H, W, Cn, K, R, S, stride)
  x = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, R, S, Cn), generator=g, device='cuda') * (R * S * Cn) ** -0.5).to(BF)
  dy = torch.randn((N, d.Ho, d.Wo, K), generator=g, device='cuda').to(BF)
  st = torch.cuda.current_stream().cuda_stream
  kpad = (K + 7) // 8 * 8
  # ---- fprop (+ fused statistics) ----
  y, stats = ops.conv_fprop(d, x, w, want_stats=(K % 8 == 0))
  yn = torch.empty_like(y)
  assert L.asm_conv2d_fprop_naive(C.byref(d), x.data_ptr(), w.data_ptr(), yn.data_ptr(), st) == 0
  _close_bf16(y, yn, tag + ' fprop vs direct', errs)
  if stats is not None:
    yf = y.float().view(-1, K)
    s = stats.double().sum(0)
    if not (torch.allclose(s[0], yf.double().sum(0), rtol=1e-4, atol=0.5) and
            torch.allclose(s[1], (yf.double() ** 2).sum(0), rtol=1e-4, atol=0.5)):
      errs.append(tag + ' fused statistics')
  # ---- dgrad ----
  if K % 8 == 0:
    wt = torch.empty((Cn, R, S, K), dtype=BF, device='cuda')
    ops.filter_transpose(w, wt, K, R, S, Cn)
    dx = ops.conv_dgrad(d, dy, wt)
    dxn = torch.empty_like(dx)
    assert L.asm_conv2d_dgrad_naive(C.byref(d), dy.data_ptr(), w.data_ptr(), dxn.data_ptr(), st) == 0
    _close_bf16(dx, dxn, tag + ' dgrad vs direct', errs)
  # ---- wgrad with the library's own plan ----
  dw = torch.empty((K, R, S, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, x, dy, dw)
  dwn = torch.empty_like(dw)
  assert L.asm_conv2d_wgrad_naive(C.byref(d), x.data_ptr(), dy.data_ptr(), dwn.data_ptr(), st) == 0
  _close_f32(dw, dwn, tag + ' wgrad vs direct (plan %s)' % _plan(L, d), errs)
  if against_oracle:
    from oracle import assembled_oracle as O
    xr = x.float().cpu().requires_grad_(True)
    wr = w.float().cpu().requires_grad_(True)
    # R == S on every generic layer; the oracle's conv takes the kernel size
    yr = O._conv_raw(xr.permute(0, 3, 1, 2), wr.permute(1, 2, 3, 0), R, stride)
    gx, gw = torch.autograd.grad(yr, [xr, wr], dy.float().cpu().permute(0, 3, 1, 2))
    _close_bf16(y.cpu(), yr.detach().permute(0, 2, 3, 1), tag + ' fprop vs oracle', errs, rel=4e-3)
    _close_bf16(yn.cpu(), yr.detach().permute(0, 2, 3, 1), tag + ' direct fprop vs oracle', errs, rel=4e-3)
    if K % 8 == 0:
      _close_bf16(dx.cpu(), gx, tag + ' dgrad vs oracle', errs, rel=4e-3)
    _close_f32(dw.cpu(), gw, tag + ' wgrad vs oracle', errs)
    _close_f32(dwn.cpu(), gw, tag + ' direct wgrad vs oracle', errs)
  return _plan(L, d)


def _run_stem(L, key, errs, against_oracle):
  """3-channel first conv over the zero-haloed [N][H+6][W+6][4] buffer (R = k, S = 1, packed rows)."""
  from assembled_cnn_amd import nn, ops
  N, Hp, Wp, _, K, ksize, _, _, _ = key
  H, W = Hp - 6, Wp - 6
  tag = 'stem N%d %dx%d k%d ->K%d' % (N, H, W, ksize, K)
  dev = torch.device('cuda')
  arena = nn.ParamArena()
  c = nn.Ctx(arena, True, True, 0.997, dev, False)
  conv = nn.ConvKernel(c, ksize, 3, K, stem=True)
  arena.finalize(dev, 0)
  g = torch.Generator(device='cuda').manual_seed(ksize)
  x = (torch.randn((N, H, W, 3), generator=g, device='cuda') * 50.0).to(BF)
  xp = ops.stem_pad_input(x)
  d = conv.desc(N, H, W, 2)
  st = torch.cuda.current_stream().cuda_stream
  y, _ = conv.fprop(d, xp, False)
  view = conv._stem_view(xp)
  yn = torch.empty_like(y)
  assert L.asm_conv2d_fprop_naive(C.byref(d), view.data_ptr(), conv.weight().data_ptr(), yn.data_ptr(), st) == 0
  _close_bf16(y, yn, tag + ' fprop vs direct', errs)
  dy = torch.randn(tuple(y.shape), generator=g, device='cuda').to(BF)
  dwp = torch.empty((K, ksize, conv.stem_len), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, view, dy, dwp)
  dwn = torch.empty_like(dwp)
  assert L.asm_conv2d_wgrad_naive(C.byref(d), view.data_ptr(), dy.data_ptr(), dwn.data_ptr(), st) == 0
  _close_f32(dwp, dwn, tag + ' wgrad vs direct (plan %s)' % _plan(L, d), errs)
  if against_oracle:
    from oracle import assembled_oracle as O
    wkr = arena.wb(conv.name).float().cpu().requires_grad_(True)
    yr = O._conv_raw(x.float().cpu().permute(0, 3, 1, 2), wkr.permute(1, 2, 3, 0), ksize, 2)
    (gw,) = torch.autograd.grad(yr, [wkr], dy.float().cpu().permute(0, 3, 1, 2))
    _close_bf16(y.cpu(), yr.detach().permute(0, 2, 3, 1), tag + ' fprop vs oracle', errs, rel=4e-3)
    conv.backward(d, xp, dy, False)
    _close_f32(arena.g(conv.name).cpu(), gw, tag + ' wgrad vs oracle', errs)


@pytest.mark.parametrize('workload', ['assemble-r50', 'r50'])
def test_every_conv_shape_at_batch_256_vs_direct(hip_lib, workload):
  """Full tensors at the benchmark batch, library-chosen plans, against the direct kernels."""
  errs, big_plans = [], 0
  keys = _shapes(workload, 256)
  assert len(keys) >= 23
  for key in keys:
    if key[8]:
      _run_stem(hip_lib, key, errs, against_oracle=False)
    else:
      plan = _run_generic(hip_lib, key, errs, against_oracle=False)
      big_plans += plan[1] == 256 or plan[1] == -2      # 256 x 256 tiles, or the resident-row kernel of the deep 3x3 layers
    torch.cuda.synchronize()
  if workload == 'assemble-r50':
    assert big_plans >= 5, 'the 256 x 256 / resident-row weight-gradient kernels should carry the heavy layers (%d did)' % big_plans
  assert not errs, '\n'.join(errs)


@pytest.mark.parametrize('workload', ['assemble-r50', 'r50'])
def test_every_conv_shape_on_an_8_image_slice_vs_oracle(hip_lib, workload):
  """The same layers at N = 8 against the CPU oracle and its autograd (also pins the direct kernels)."""
  errs = []
  for key in _shapes(workload, 8):
    if key[8]:
      _run_stem(hip_lib, key, errs, against_oracle=True)
    else:
      _run_generic(hip_lib, key, errs, against_oracle=True)
  assert not errs, '\n'.join(errs)


BIG_WGRAD_SHAPES = [
    # N, H,  W,  C,   K,   k, stride        what it exercises in wgrad8_kernel (the 256 x 256 tile)
    (2, 14, 14, 128, 256, 3, 1),     # 1152 columns -> 5 column tiles, the last one half empty; 392 pixels (ragged steps)
    (3, 7, 7, 256, 512, 1, 1),       # 256 columns exactly, two dy row-tiles, 147 pixels
    (2, 14, 14, 256, 256, 3, 2),     # strided gather, 98 pixels (< 2 steps)
    (2, 16, 16, 64, 256, 3, 1),      # 576 columns -> 3 tiles with a 64-column tail
    (5, 9, 9, 72, 264, 3, 1),        # K = 264: second row-tile has 8 valid rows; channel count not a power of two
]


@pytest.mark.parametrize('splits', [0, 3], ids=['auto-splits', 'forced-3-splits'])
@pytest.mark.parametrize('shape', BIG_WGRAD_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_wgrad_256x256_kernel_forced_vs_oracle(hip_lib, shape, splits, monkeypatch):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  util.set_knob(monkeypatch, 'ASM_WGRAD_BIG', '1')
  if splits:
    util.set_knob(monkeypatch, 'ASM_WGRAD_SPLITS', str(splits))
  N, H, W, Cn, K, k, stride = shape
  g = torch.Generator().manual_seed(41)
  x = torch.randn((N, H, W, Cn), generator=g).to(BF)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  dy = torch.randn((N, d.Ho, d.Wo, K), generator=g).to(BF)
  plan = _plan(hip_lib, d)
  assert plan[0] == 256 and plan[1] == 256, 'ASM_WGRAD_BIG=1 must select the 256 x 256 kernel: %s' % plan
  if splits:
    assert plan[4] > 1, plan
  dw = torch.empty((K, k, k, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw)
  wr = torch.zeros((K, k, k, Cn), requires_grad=True)
  yr = O._conv_raw(x.float().permute(0, 3, 1, 2), wr.permute(1, 2, 3, 0), k, stride)
  (gw,) = torch.autograd.grad(yr, [wr], dy.float().permute(0, 3, 1, 2))
  r = util.rel_l2(dw.cpu(), gw)
  assert r <= 2e-3, 'wgrad8 rel_l2 %.3e (plan %s)' % (r, plan)
  # the same call with the 128-wide kernels gives the same numbers up to summation order
  util.set_knob(monkeypatch, 'ASM_WGRAD_BIG', '0')
  dw0 = torch.empty_like(dw)
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw0)
  assert _plan(hip_lib, d)[1] == 128
  assert util.rel_l2(dw, dw0) <= 1e-4
  # deterministic (fixed-order slab reduce)
  util.set_knob(monkeypatch, 'ASM_WGRAD_BIG', '1')
  dw1 = torch.empty_like(dw)
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw1)
  assert torch.equal(dw, dw1)


def test_wgrad_256x256_kernel_on_its_own_plan_vs_direct(hip_lib):
  """N = 32, 14 x 14, 512 -> 1024, 3 x 3 = 59 GFLOP: above the 40 GFLOP threshold, so the library picks the
  8-wave kernel and a multi-split plan by itself."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, k = 32, 14, 14, 512, 1024, 3
  g = torch.Generator(device='cuda').manual_seed(5)
  x = torch.randn((N, H, W, Cn), generator=g, device='cuda').to(BF)
  dy = torch.randn((N, H, W, K), generator=g, device='cuda').to(BF)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, 1)
  plan = _plan(hip_lib, d)
  assert plan[:2] == [256, 256], plan
  dw = torch.empty((K, k, k, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, x, dy, dw)
  dwn = torch.empty_like(dw)
  st = torch.cuda.current_stream().cuda_stream
  assert hip_lib.asm_conv2d_wgrad_naive(C.byref(d), x.data_ptr(), dy.data_ptr(), dwn.data_ptr(), st) == 0
  assert util.rel_l2(dw, dwn) <= 2e-3


@pytest.mark.parametrize('rows', [256, 1024, 4096])
def test_bn_partial_row_cap_is_a_per_call_knob(hip_lib, rows, monkeypatch):
  """ASM_BN_ROWS is read on every call: the same reduction through three partial-row tilings."""
  from assembled_cnn_amd import ops
  util.set_knob(monkeypatch, 'ASM_BN_ROWS', str(rows))
  M, Cn = 50176, 64
  g = torch.Generator(device='cuda').manual_seed(9)
  x = (torch.randn((M, Cn), generator=g, device='cuda') * 2 + 0.5).to(BF)
  part = ops.bn_stats(x, M, Cn)
  assert part.shape[0] <= max(rows, 1) + 1
  s = part.double().sum(0)
  assert torch.allclose(s[0], x.double().sum(0), rtol=1e-5, atol=1e-2)
  assert torch.allclose(s[1], (x.double() ** 2).sum(0), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize('v2,parity', [(0, 1), (1, 0), (1, 1)], ids=['general-kernel', 'generic-s2-dgrad', 'parity-classes'])
def test_igemm_variants_are_per_call_knobs(hip_lib, v2, parity, monkeypatch):
  """ASM_IGEMM_MODE / ASM_DGRAD_PARITY flipped inside one process give identical bits (same accumulation order)."""
  from assembled_cnn_amd import ops
  N, H, W, Cn, K, k, stride = 4, 14, 14, 128, 256, 3, 2
  g = torch.Generator(device='cuda').manual_seed(3)
  dy = torch.randn((N, 7, 7, K), generator=g, device='cuda').to(BF)
  w = (torch.randn((K, k, k, Cn), generator=g, device='cuda') * (9 * Cn) ** -0.5).to(BF)
  wt = torch.empty((Cn, k, k, K), dtype=BF, device='cuda')
  ops.filter_transpose(w, wt, K, k, k, Cn)
  d = ops.make_conv_desc(N, H, W, Cn, K, k, k, stride)
  ref = ops.conv_dgrad(d, dy, wt)
  util.set_knob(monkeypatch, 'ASM_IGEMM_MODE', str(1 - v2))
  util.set_knob(monkeypatch, 'ASM_DGRAD_PARITY', str(parity))
  out = ops.conv_dgrad(d, dy, wt)
  dxn = torch.empty_like(ref)
  st = torch.cuda.current_stream().cuda_stream
  assert hip_lib.asm_conv2d_dgrad_naive(C.byref(d), dy.data_ptr(), w.data_ptr(), dxn.data_ptr(), st) == 0
  assert util.rel_l2(out.float(), dxn.float()) <= 3e-3
  assert util.rel_l2(out.float(), ref.float()) <= 2e-3


@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 32), (3, 8, 16, 32, 64), (1, 24, 48, 32, 32), (2, 16, 16, 64, 64),
                                   (8, 112, 112, 64, 32),
                                   # whole-row bands of 112 pixels (2 x 56, 4 x 28); K = 128 as two 64-channel halves
                                   (3, 56, 56, 64, 128), (5, 28, 28, 64, 128), (2, 56, 56, 32, 64), (3, 28, 28, 64, 64),
                                   (2, 28, 28, 32, 32), (1, 4, 28, 64, 128), (1, 2, 56, 32, 128)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_wgrad_halo_kernel_forced_vs_oracle(hip_lib, shape, monkeypatch):
  """wgrad_halo_kernel (persistent, halo-resident, all nine taps per patch) on small shapes and on an 8-image slice of
  its own layer: against the oracle's autograd and bit-reproducible."""
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  util.set_knob(monkeypatch, 'ASM_WGRAD_HALO', '2')
  N, H, W, Cn, K = shape
  g = torch.Generator().manual_seed(43)
  x = torch.randn((N, H, W, Cn), generator=g).to(BF)
  dy = torch.randn((N, H, W, K), generator=g).to(BF)
  d = ops.make_conv_desc(N, H, W, Cn, K, 3, 3, 1)
  plan = _plan(hip_lib, d)
  assert plan[1] == -1, 'the halo form should be selected: %s' % plan
  dw = torch.empty((K, 3, 3, Cn), dtype=torch.float32, device='cuda')
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw)
  wr = torch.zeros((K, 3, 3, Cn), requires_grad=True)
  yr = O._conv_raw(x.float().permute(0, 3, 1, 2), wr.permute(1, 2, 3, 0), 3, 1)
  (gw,) = torch.autograd.grad(yr, [wr], dy.float().permute(0, 3, 1, 2))
  r = util.rel_l2(dw.cpu(), gw)
  assert r <= 2e-3, 'wgrad_halo rel_l2 %.3e (plan %s)' % (r, plan)
  dw1 = torch.empty_like(dw)
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw1)
  assert torch.equal(dw, dw1)
  util.set_knob(monkeypatch, 'ASM_WGRAD_HALO', '0')
  dw0 = torch.empty_like(dw)
  ops.conv_wgrad(d, x.cuda(), dy.cuda(), dw0)
  assert _plan(hip_lib, d)[1] != -1 and util.rel_l2(dw, dw0) <= 1e-4
