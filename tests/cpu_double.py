"""A CPU *test double* of the C ABI in include/asm_hip.h (TEST INFRASTRUCTURE ONLY).

It lets the GPU-less test tier drive the real host code of ``assembled_cnn_amd`` (ops.py, nn.py,
model.py, train.py, dp.py: parameter arenas, the backward tape, the topology walker, the optimiser
wiring) end to end and compare it with the oracle.  It is installed with
``assembled_cnn_amd.ops.set_library(CpuDouble(), is_double=True)`` by tests only; the product never
imports it and has no CPU path.

Every function follows the semantics written in the header, computing in float32 with torch CPU ops
and rounding to bf16 exactly where the kernels store bf16.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

_CT = {'bf16': (C.c_int16, torch.bfloat16), 'f32': (C.c_float, torch.float32), 'u8': (C.c_uint8, torch.uint8),
       'i32': (C.c_int32, torch.int32)}


def T(ptr, shape, kind):
  """Tensor aliasing the caller's memory."""
  if not ptr:
    return None
  n = int(np.prod(shape))
  ct, tt = _CT[kind]
  arr = np.ctypeslib.as_array((ct * n).from_address(ptr))
  t = torch.from_numpy(arr)
  if kind == 'bf16':
    t = t.view(torch.bfloat16)
  return t.view(tuple(shape))


def _desc(d):
  return d._obj if hasattr(d, '_obj') else d


def _pitches(d):
  img = d.x_img_pitch if d.x_img_pitch else d.H * d.W * d.C
  row = d.x_row_pitch if d.x_row_pitch else d.W * d.C
  pix = d.x_pix_pitch if d.x_pix_pitch else d.C
  return int(img), int(row), int(pix)


def _gather(xptr, d, r, s):
  """[N,Ho,Wo,C] float32 of x(n, ho*stride+r-pad, wo*stride+s-pad, :) with zeros out of range."""
  img, row, pix = _pitches(d)
  ih = torch.arange(d.Ho) * d.stride + r - d.pad
  iw = torch.arange(d.Wo) * d.stride + s - d.pad
  vh = (ih >= 0) & (ih < d.H)
  vw = (iw >= 0) & (iw < d.W)
  valid = vh[:, None] & vw[None, :]
  base = (torch.arange(d.N)[:, None, None] * img + ih.clamp(0, d.H - 1)[None, :, None] * row +
          iw.clamp(0, d.W - 1)[None, None, :] * pix)
  idx = base[..., None] + torch.arange(d.C)
  idx = torch.where(valid[None, :, :, None].expand_as(idx), idx, torch.zeros_like(idx))
  n = int(idx.max()) + 1
  flat = T(xptr, (n,), 'bf16').float()
  g = flat[idx.reshape(-1)].view(d.N, d.Ho, d.Wo, d.C)
  return g * valid[None, :, :, None]


class CpuDouble(object):
  def __init__(self):
    self._err = b''

  # ---- plumbing ------------------------------------------------------------------------------------
  def asm_last_error(self):
    return self._err

  def asm_launch_count(self):
    return 0

  # launch tape: the double records nothing (it has no launches to replay); it keeps the SEGMENT bookkeeping, which is
  # what the host logic above it depends on (dp.GradSync cuts a recording at its bucket launches)
  def asm_stream_join(self, dst, src):
    return 0

  def asm_tape_begin(self):
    if getattr(self, '_tape_open', False):
      self._err = b'tape_begin: this thread is already recording a tape'
      return -1
    self._tape_open, self._tape_marks = True, 0
    self._tapes = getattr(self, '_tapes', 0) + 1
    return self._tapes

  def asm_tape_mark(self):
    if not getattr(self, '_tape_open', False):
      self._err = b'tape_mark: no tape is being recorded by this thread'
      return -1
    self._tape_marks += 1
    return self._tape_marks

  def asm_tape_end(self):
    if not getattr(self, '_tape_open', False):
      self._err = b'tape_end: no tape is being recorded by this thread'
      return -1
    self._tape_open = False
    return self._tapes

  def asm_tape_info(self, tape, info):
    arr = info._obj if hasattr(info, '_obj') else info
    for i, v in enumerate((0, 0, 0, 0, self._tape_marks + 1, 0)):
      arr[i] = v
    return 0

  def asm_tape_replay(self, tape, segment):
    self._err = b'tape_replay: the CPU test double records no launches'
    return -2

  def asm_tape_free(self, tape):
    return 0

  def asm_abi_version(self):
    return 1

  def asm_tuning_defaults(self, t):
    return None

  def asm_set_tuning(self, t):
    return 0

  def asm_get_tuning(self, t):
    return None

  # ---- conv ----------------------------------------------------------------------------------------
  def asm_conv2d_stats_blocks(self, d):
    d = _desc(d)
    return (d.N * d.Ho * d.Wo + 127) // 128

  @staticmethod
  def _validate(d, what):
    """the argument checks of csrc/conv_igemm.hip / conv_wgrad.hip, so host-side misuse fails on CPU too"""
    assert d.C % 8 == 0 and d.stride in (1, 2) and d.Ho > 0 and d.Wo > 0 and 0 <= d.pad < 64, what
    ldy = d.ldy if d.ldy else d.K
    if what == 'fprop':
      assert ldy % (4 if d.out_f32 else 8) == 0 and ldy >= d.K, 'fprop: bad ldy'
    elif what == 'dgrad':
      assert d.K % 8 == 0 and not d.out_f32 and not d.x_img_pitch and not d.x_row_pitch and not d.x_pix_pitch, \
          'dgrad: custom pitches / f32 output / unpadded K not supported'
    else:
      assert ldy % 8 == 0 and ldy >= d.K and not d.out_f32, 'wgrad: bad dy row stride'

  def asm_conv2d_fprop(self, d, x, w, y, stats, stream):
    d = _desc(d)
    self._validate(d, 'fprop')
    assert not (stats and d.out_f32)
    ldy = d.ldy if d.ldy else d.K
    wt = T(w, (d.K, d.R, d.S, d.C), 'bf16').float()
    acc = torch.zeros(d.N, d.Ho, d.Wo, d.K)
    for r in range(d.R):
      for s in range(d.S):
        acc += _gather(x, d, r, s) @ wt[:, r, s, :].t()
    out = T(y, (d.N * d.Ho * d.Wo, ldy), 'f32' if d.out_f32 else 'bf16')
    out[:, :d.K] = acc.view(-1, d.K).to(out.dtype)
    if stats:
      M = d.N * d.Ho * d.Wo
      nb = (M + 127) // 128
      st = T(stats, (nb, 2, d.K), 'f32')
      v = out[:, :d.K].float()
      for b in range(nb):
        blk = v[b * 128:(b + 1) * 128]
        st[b, 0] = blk.sum(0)
        st[b, 1] = (blk * blk).sum(0)
    return 0

  def asm_conv2d_fprop_bn(self, d, x, w, y, scale, shift, residual, relu, stream):
    d = _desc(d)
    self._validate(d, 'fprop')
    if d.out_f32 or d.K % 8 or not scale or not shift or (d.ldy and d.ldy != d.K):
      self._err = b'conv fprop_bn: bad arguments'
      return -1
    wt = T(w, (d.K, d.R, d.S, d.C), 'bf16').float()
    acc = torch.zeros(d.N, d.Ho, d.Wo, d.K)
    for r in range(d.R):
      for s in range(d.S):
        acc += _gather(x, d, r, s) @ wt[:, r, s, :].t()
    acc = acc.view(-1, d.K).to(torch.bfloat16).float()      # the kernel transposes the tile through LDS as bf16
    out = acc * T(scale, (d.K,), 'f32') + T(shift, (d.K,), 'f32')
    if residual:
      out = out + T(residual, (d.N * d.Ho * d.Wo, d.K), 'bf16').float()
    if relu:
      out = out.clamp(min=0)
    T(y, (d.N * d.Ho * d.Wo, d.K), 'bf16').copy_(out)
    return 0

  def asm_conv2d_dgrad(self, d, dy, wt, addend, dx, stream, addend_mask=0):
    d = _desc(d)
    self._validate(d, 'dgrad')
    g = T(dy, (d.N, d.Ho, d.Wo, d.K), 'bf16').float()
    w_crsk = T(wt, (d.C, d.R, d.S, d.K), 'bf16').float()
    w_oihw = w_crsk.permute(3, 0, 1, 2)  # [K, C, R, S]
    x = torch.zeros(d.N, d.C, d.H, d.W, requires_grad=True)
    pad_after_h = max((d.Ho - 1) * d.stride + d.R - d.pad - d.H, 0)
    pad_after_w = max((d.Wo - 1) * d.stride + d.S - d.pad - d.W, 0)
    yy = F.conv2d(F.pad(x, (d.pad, pad_after_w, d.pad, pad_after_h)), w_oihw, stride=d.stride)
    yy = yy[:, :, :d.Ho, :d.Wo]
    (gx,) = torch.autograd.grad(yy, x, g.permute(0, 3, 1, 2))
    gx = gx.permute(0, 2, 3, 1).to(torch.bfloat16).float()
    if addend:
      ad = T(addend, (d.N, d.H, d.W, d.C), 'bf16').float()
      if addend_mask:
        ad = ad * self._unpack_mask(addend_mask, d.N * d.H * d.W, d.C).view(ad.shape)
      gx = gx + ad
    T(dx, (d.N, d.H, d.W, d.C), 'bf16').copy_(gx)
    return 0

  def asm_conv2d_dgrad_pooled(self, d, dy, wt, addend, addend_mask, pool_dy, pk, pst, ppad, pHo, pWo, cv, dx, stream):
    dd = _desc(d)
    if not (dd.R == 1 and dd.S == 1 and dd.stride == 1 and dd.pad == 0):
      self._err = b'conv dgrad_pooled: needs a 1x1 stride-1 convolution'
      return -2
    if addend_mask:
      rc = self.asm_conv2d_dgrad_masked(d, dy, wt, addend, addend_mask, dx, stream)
    else:
      rc = self.asm_conv2d_dgrad(d, dy, wt, addend, dx, stream)
    if rc:
      return rc
    return self.asm_avgpool_bwd(pool_dy, dx, dd.N, dd.H, dd.W, dd.C, pk, pst, ppad, pHo, pWo, cv, dx, stream)

  def asm_conv2d_dgrad_masked(self, d, dy, wt, addend, addend_mask, dx, stream):
    return self.asm_conv2d_dgrad(d, dy, wt, addend, dx, stream, addend_mask)

  @staticmethod
  def _bnred_covers(d):
    """the layers asm_conv2d_dgrad_bnred covers on the GPU side, as far as a host test can tell: every 1x1 stride-1 layer, and the
    3x3 stride-1 pad-1 layers with K % 64 == 0 and more than 64 input channels (igemm3 / igemm8 territory)"""
    if d.stride != 1 or d.C % 8:
      return False
    if d.R == 1 and d.S == 1 and d.pad == 0:
      return True
    if not (d.R == 3 and d.S == 3 and d.pad == 1):
      return False
    if d.C in (32, 64) and d.K in (32, 64) and d.H % 8 == 0 and d.W % 16 == 0:      # conv_halo_kernel: not there
      return False
    return d.K % 64 == 0 and d.C > 64

  def asm_conv2d_dgrad_bnred_blocks(self, d):
    d = _desc(d)
    return (d.N * d.H * d.W + 127) // 128 if self._bnred_covers(d) else 0

  def asm_conv2d_dgrad_bnred(self, d, dy, wt, addend, addend_mask, bn_y, bn_mask, partial, dx, stream):
    dd = _desc(d)
    if not self._bnred_covers(dd):
      self._err = b'conv dgrad_bnred: layer not covered'
      return -2
    rc = self.asm_conv2d_dgrad(d, dy, wt, addend, dx, stream, addend_mask) if addend_mask else \
        self.asm_conv2d_dgrad(d, dy, wt, addend, dx, stream)
    if rc:
      return rc
    M, Cn = dd.N * dd.H * dd.W, dd.C
    dz = T(dx, (M, Cn), 'bf16').float()
    if bn_mask:
      dz = dz * self._unpack_mask(bn_mask, M, Cn)
    y = T(bn_y, (M, Cn), 'bf16').float()
    blocks = (M + 127) // 128
    out = T(partial, (blocks, 2, Cn), 'f32')
    for b in range(blocks):
      out[b, 0] = dz[b * 128:(b + 1) * 128].sum(0)
      out[b, 1] = (dz[b * 128:(b + 1) * 128] * y[b * 128:(b + 1) * 128]).sum(0)
    return 0

  @staticmethod
  def _unpack_mask(mask, M, Cn):
    mk = T(mask, (M, Cn // 8), 'u8').to(torch.int32)
    bits = (mk[:, :, None] >> torch.arange(8, dtype=torch.int32)) & 1
    return bits.reshape(M, Cn).float()

  def asm_mask_apply(self, dy, mask, dx, n, stream):
    T(dx, (n,), 'bf16').copy_(T(dy, (n,), 'bf16').float() * self._unpack_mask(mask, n // 8, 8).view(n))
    return 0

  def asm_conv2d_wgrad_workspace_bytes(self, d):
    return 64

  def asm_conv2d_wgrad(self, d, x, dy, dw, ws, ws_bytes, stream):
    d = _desc(d)
    self._validate(d, 'wgrad')
    ldy = d.ldy if d.ldy else d.K
    g = T(dy, (d.N * d.Ho * d.Wo, ldy), 'bf16').float()[:, :d.K]
    out = T(dw, (d.K, d.R, d.S, d.C), 'f32')
    for r in range(d.R):
      for s in range(d.S):
        out[:, r, s, :] = g.t() @ _gather(x, d, r, s).view(-1, d.C)
    return 0

  def asm_filter_transpose(self, w, wt, K, R, S, Cn, ldk, stream):
    ldk = ldk if ldk else K
    src = T(w, (K, R, S, Cn), 'bf16')
    dst = T(wt, (Cn, R, S, ldk), 'bf16')
    dst[..., :K] = src.permute(3, 1, 2, 0)
    return 0

  def asm_filter_transpose_batched(self, w, wt, table, nl, total, stream):
    tab = T(table, (nl, 8), 'i32')
    for l in range(nl):
      so, do, K, RS, Cn, ldk, eb, _ = [int(v) for v in tab[l]]
      src = T(w + 2 * so, (K, RS, Cn), 'bf16')
      dst = T(wt + 2 * do, (Cn, RS, ldk), 'bf16')
      dst[..., :K] = src.permute(2, 1, 0)
    return 0

  def asm_filter_transpose_tiled(self, w, wt, table, nl, total_tiles, stream):
    tab = T(table, (nl, 8), 'i32')
    for l in range(nl):
      so, do, K, RS, Cn, ldk, _, _ = [int(v) for v in tab[l]]
      src = T(w + 2 * so, (K, RS, Cn), 'bf16')
      dst = T(wt + 2 * do, (Cn, RS, ldk), 'bf16')
      dst.zero_()
      dst[..., :K] = src.permute(2, 1, 0)
    return 0

  def asm_stem_pack_filter(self, w, wp, K, ks, stream):
    Lr = (4 * ks + 7) // 8 * 8
    src = T(w, (K, ks, ks, 3), 'f32')
    dst = T(wp, (K, ks, Lr), 'bf16')
    dst.zero_()
    tmp = torch.zeros(K, ks, Lr // 4, 4)
    tmp[:, :, :ks, :3] = src
    dst.copy_(tmp.view(K, ks, Lr))
    return 0

  def asm_stem_unpack_grad(self, dwp, dw, K, ks, stream):
    Lr = (4 * ks + 7) // 8 * 8
    src = T(dwp, (K, ks, Lr // 4, 4), 'f32')
    T(dw, (K, ks, ks, 3), 'f32').copy_(src[:, :, :ks, :3])
    return 0

  def asm_stem_pad_input(self, x, is_f32, xp, N, H, W, stream):
    src = T(x, (N, H, W, 3), 'f32' if is_f32 else 'bf16').float()
    dst = T(xp, (N, H + 6, W + 6, 4), 'bf16')
    dst.zero_()
    dst[:, 3:3 + H, 3:3 + W, :3] = src.to(torch.bfloat16)
    return 0

  # ---- BN ------------------------------------------------------------------------------------------
  def asm_bn_stats_blocks(self, M, Cn):
    return (M + 255) // 256

  def asm_bn_stats(self, x, M, Cn, part, stream):
    v = T(x, (M, Cn), 'bf16').float()
    nb = (M + 255) // 256
    st = T(part, (nb, 2, Cn), 'f32')
    for b in range(nb):
      blk = v[b * 256:(b + 1) * 256]
      st[b, 0] = blk.sum(0)
      st[b, 1] = (blk * blk).sum(0)
    return 0

  def asm_bn_apply2(self, xa, xb, y, M, Cn, sa, ha, sb, hb, relu, mask, stream):
    zb = torch.empty((M, Cn), dtype=torch.bfloat16)
    rc = self.asm_bn_apply(xb, zb.data_ptr(), M, Cn, sb, hb, None, 0, 0, 0, 0, None, stream)
    if rc:
      return rc
    return self.asm_bn_apply(xa, y, M, Cn, sa, ha, zb.data_ptr(), 1, relu, 0, 0, mask, stream)

  def asm_bn_bwd_reduce2(self, dy, xa, xb, mask, M, Cn, mean_a, invstd_a, mean_b, invstd_b, pa, pb, stream):
    for x, mean, invstd, part in ((xa, mean_a, invstd_a, pa), (xb, mean_b, invstd_b, pb)):
      rc = self.asm_bn_bwd_reduce(dy, x, mask, 2, M, Cn, mean, invstd, part, stream)
      if rc:
        return rc
    return 0

  def asm_bn_bwd_apply2(self, dy, xa, xb, mask, M, Cn, coef6, dxa, dxb, stream):
    f = C.sizeof(C.c_float)
    for i, (x, dx) in enumerate(((xa, dxa), (xb, dxb))):
      base = coef6 + 3 * i * Cn * f
      rc = self.asm_bn_bwd_apply(dy, x, mask, 2, M, Cn, base, base + Cn * f, base + 2 * Cn * f, dx, None, stream)
      if rc:
        return rc
    return 0

  def asm_bn_small_max_rows(self):
    return 4096

  def asm_bn_small_fwd(self, x, y, M, Cn, gamma, beta, eps, momentum, mm, mv, mean, invstd, relu, mask, stream):
    if M <= 0 or M > 4096 or Cn % 8:
      self._err = b'bn_small_fwd: bad arguments'
      return -1
    v = T(x, (M, Cn), 'bf16').double()
    mu = v.mean(0)
    var = (v * v).mean(0) - mu * mu
    var = var.clamp(min=0)
    isd = (1.0 / torch.sqrt(var + eps)).float()
    T(mean, (Cn,), 'f32').copy_(mu.float())
    T(invstd, (Cn,), 'f32').copy_(isd)
    g, b = T(gamma, (Cn,), 'f32'), T(beta, (Cn,), 'f32')
    sc = g * isd
    sh = b - mu.float() * sc
    if mm:
      m1, v1 = T(mm, (Cn,), 'f32'), T(mv, (Cn,), 'f32')
      unb = (var * (M / max(M - 1, 1))).float()
      m1.mul_(momentum).add_(mu.float() * (1 - momentum))
      v1.mul_(momentum).add_(unb * (1 - momentum))
    out = T(x, (M, Cn), 'bf16').float() * sc + sh
    if relu:
      if mask:
        bits = (out > 0).view(M, Cn // 8, 8).to(torch.int32)
        T(mask, (M, Cn // 8), 'u8').copy_((bits << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8))
      out = out.clamp(min=0)
    T(y, (M, Cn), 'bf16').copy_(out)
    return 0

  def asm_bn_small_bwd(self, dy, x, mask, M, Cn, gamma, mean, invstd, dgamma, dbeta, dx, stream):
    g = T(dy, (M, Cn), 'bf16').float()
    xv = T(x, (M, Cn), 'bf16').float()
    if mask:
      mk = T(mask, (M, Cn // 8), 'u8').to(torch.int32)
      bits = ((mk[..., None] >> torch.arange(8, dtype=torch.int32)) & 1).view(M, Cn).bool()
      g = torch.where(bits, g, torch.zeros_like(g))
    mu, isd, gm = T(mean, (Cn,), 'f32'), T(invstd, (Cn,), 'f32'), T(gamma, (Cn,), 'f32')
    xhat = (xv - mu) * isd
    db = g.double().sum(0)
    dg = (g * xhat).double().sum(0)
    T(dbeta, (Cn,), 'f32').copy_(db.float())
    T(dgamma, (Cn,), 'f32').copy_(dg.float())
    A = (gm * isd).double()
    B = -(gm * isd * isd).double() * dg / M
    Cc = -(gm * isd).double() * db / M - B * mu.double()
    T(dx, (M, Cn), 'bf16').copy_((A.float() * g + B.float() * xv + Cc.float()))
    return 0

  # ---- small dense layers (csrc/dense_small.hip) ----------------------------------------------------
  def asm_dense_small(self, p, ldp, q, ldq, M, N, K, out, ldo, out_f32, addend, stream):
    if K % 16 or ldp % 8 or ldq % 8 or ldp < K or ldq < K or ldo < N or ldo % 4:
      self._err = b'dense_small: bad arguments'
      return -1
    pv = T(p, (M, ldp), 'bf16').float()[:, :K]
    qv = T(q, (N, ldq), 'bf16').float()[:, :K]
    acc = pv @ qv.t()
    if addend:
      acc = acc + T(addend, (M, ldo), 'bf16').float()[:, :N]
    o = T(out, (M, ldo), 'f32' if out_f32 else 'bf16')
    o[:, :N] = acc.to(o.dtype)
    o[:, N:] = 0          # pad columns of a padded output row are written as zeros (include/asm_hip.h)
    return 0

  def asm_dense_small_wgrad(self, x, ldx, dy, ldy, M, Cin, Cout, dw, ldw, stream):
    if ldx < Cin or ldy < Cout or ldw < Cin or ldw % 4:
      self._err = b'dense_small_wgrad: bad row strides'
      return -1
    xv = T(x, (M, ldx), 'bf16').float()[:, :Cin]
    gv = T(dy, (M, ldy), 'bf16').float()[:, :Cout]
    T(dw, (Cout, ldw), 'f32')[:, :Cin] = gv.t() @ xv
    return 0

  def asm_bn_partials_compact(self, part, blocks, Cn, out, groups, stream):
    per = -(-blocks // groups)
    assert -(-blocks // per) == groups
    src = T(part, (blocks, 2, Cn), 'f32')
    dst = T(out, (groups, 2, Cn), 'f32')
    for g in range(groups):
      dst[g] = src[g * per:(g + 1) * per].double().sum(0).float()
    return 0

  def asm_bn_finalize(self, part, blocks, M, Cn, gamma, beta, eps, momentum, mm, mv, mean, invstd, scale, shift,
                      stream):
    st = T(part, (blocks, 2, Cn), 'f32').double().sum(0)
    mu = st[0] / M
    var = (st[1] / M - mu * mu).clamp(min=0)
    inv = 1.0 / torch.sqrt(var + eps)
    g, b = T(gamma, (Cn,), 'f32'), T(beta, (Cn,), 'f32')
    T(mean, (Cn,), 'f32').copy_(mu.float())
    T(invstd, (Cn,), 'f32').copy_(inv.float())
    sc = g * inv.float()
    T(scale, (Cn,), 'f32').copy_(sc)
    T(shift, (Cn,), 'f32').copy_(b - mu.float() * sc)
    if mm:
      tm, tv = T(mm, (Cn,), 'f32'), T(mv, (Cn,), 'f32')
      unb = var * (M / max(M - 1, 1))
      tm.copy_(tm * momentum + mu.float() * (1 - momentum))
      tv.copy_(tv * momentum + unb.float() * (1 - momentum))
    return 0

  def asm_bn_infer_coeffs(self, Cn, gamma, beta, mm, mv, eps, scale, shift, stream):
    g, b = T(gamma, (Cn,), 'f32'), T(beta, (Cn,), 'f32')
    inv = 1.0 / torch.sqrt(T(mv, (Cn,), 'f32') + eps)
    sc = g * inv
    T(scale, (Cn,), 'f32').copy_(sc)
    T(shift, (Cn,), 'f32').copy_(b - T(mm, (Cn,), 'f32') * sc)
    return 0

  def asm_bn_apply(self, x, y, M, Cn, scale, shift, residual, res_mode, relu, H, W, mask_out, stream):
    v = T(x, (M, Cn), 'bf16').float() * T(scale, (Cn,), 'f32') + T(shift, (Cn,), 'f32')
    if res_mode == 1:
      v = v + T(residual, (M, Cn), 'bf16').float()
    elif res_mode == 2:
      N = M // (H * W)
      r = T(residual, (N, H // 2, W // 2, Cn), 'bf16').float()
      r = r.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
      v = v + r.reshape(M, Cn)
    if relu:
      if mask_out:
        bits = (v > 0).view(M, Cn // 8, 8).to(torch.int32)
        packed = (bits << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8)
        T(mask_out, (M, Cn // 8), 'u8').copy_(packed)
      v = F.relu(v)
    T(y, (M, Cn), 'bf16').copy_(v)
    return 0

  def _dz(self, dy, yout, relu, M, Cn):
    g = T(dy, (M, Cn), 'bf16').float()
    if relu == 1:
      g = g * (T(yout, (M, Cn), 'bf16').float() > 0)
    elif relu == 2:
      mk = T(yout, (M, Cn // 8), 'u8').to(torch.int32)
      bits = ((mk[:, :, None] >> torch.arange(8, dtype=torch.int32)) & 1).view(M, Cn)
      g = g * bits
    return g

  def asm_bn_bwd_reduce(self, dy, x, yout, relu, M, Cn, mean, invstd, part, stream):
    g = self._dz(dy, yout, relu, M, Cn)
    xh = (T(x, (M, Cn), 'bf16').float() - T(mean, (Cn,), 'f32')) * T(invstd, (Cn,), 'f32')
    nb = (M + 255) // 256
    st = T(part, (nb, 2, Cn), 'f32')
    for b in range(nb):
      sl = slice(b * 256, (b + 1) * 256)
      st[b, 0] = g[sl].sum(0)
      st[b, 1] = (g[sl] * xh[sl]).sum(0)
    return 0

  def asm_bn_bwd_finalize(self, part, blocks, M, Cn, gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, stream):
    st = T(part, (blocks, 2, Cn), 'f32').double().sum(0)
    db, dg = st[0], st[1]
    g, mu, inv = (T(p, (Cn,), 'f32').double() for p in (gamma, mean, invstd))
    T(dbeta, (Cn,), 'f32').copy_(db.float())
    T(dgamma, (Cn,), 'f32').copy_(dg.float())
    A = g * inv
    B = -g * inv * inv * dg / M
    Cc = -g * inv * db / M - B * mu
    T(cA, (Cn,), 'f32').copy_(A.float())
    T(cB, (Cn,), 'f32').copy_(B.float())
    T(cC, (Cn,), 'f32').copy_(Cc.float())
    return 0

  def asm_bn_bwd_finalize_raw(self, part, blocks, M, Cn, gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, stream):
    st = T(part, (blocks, 2, Cn), 'f32').double().sum(0)
    mu, inv = (T(p, (Cn,), 'f32').double() for p in (mean, invstd))
    fixed = torch.stack([st[0], inv * (st[1] - mu * st[0])]).float().reshape(1, 2, Cn).contiguous()
    return self.asm_bn_bwd_finalize(fixed.data_ptr(), 1, M, Cn, gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, stream)

  def asm_bn_bwd_apply(self, dy, x, yout, relu, M, Cn, cA, cB, cC, dx, dz_out, stream):
    g = self._dz(dy, yout, relu, M, Cn)
    if dz_out:
      T(dz_out, (M, Cn), 'bf16').copy_(g)
    v = T(cA, (Cn,), 'f32') * g + T(cB, (Cn,), 'f32') * T(x, (M, Cn), 'bf16').float() + T(cC, (Cn,), 'f32')
    T(dx, (M, Cn), 'bf16').copy_(v)
    return 0

  # ---- pools ---------------------------------------------------------------------------------------
  @staticmethod
  def _nchw(ptr, N, H, W, Cn, grad=False):
    t = T(ptr, (N, H, W, Cn), 'bf16').float().permute(0, 3, 1, 2).contiguous()
    return t.requires_grad_(True) if grad else t

  def _maxpool(self, x):
    H, W = x.shape[2], x.shape[3]
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    ph = max((Ho - 1) * 2 + 3 - H, 0)
    pw = max((Wo - 1) * 2 + 3 - W, 0)
    xp = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float('-inf'))
    return F.max_pool2d(xp, 3, 2)

  def asm_maxpool3x3s2_fwd(self, x, y, am, N, H, W, Cn, stream):
    xt = self._nchw(x, N, H, W, Cn)
    out = self._maxpool(xt)
    T(y, (N, out.shape[2], out.shape[3], Cn), 'bf16').copy_(out.permute(0, 2, 3, 1))
    self._mp_x = xt  # the double keeps the input instead of decoding the argmax codes
    return 0

  def asm_maxpool3x3s2_bwd(self, dy, am, dx, N, H, W, Cn, stream):
    xt = self._mp_x.clone().requires_grad_(True)
    out = self._maxpool(xt)
    g = T(dy, (N, out.shape[2], out.shape[3], Cn), 'bf16').float().permute(0, 3, 1, 2)
    (gx,) = torch.autograd.grad(out, xt, g)
    T(dx, (N, H, W, Cn), 'bf16').copy_(gx.permute(0, 2, 3, 1))
    return 0

  @staticmethod
  def _avgpool(x, k, stride, pad, Ho, Wo, count_valid):
    H, W = x.shape[2], x.shape[3]
    pa_h = max((Ho - 1) * stride + k - pad - H, 0)
    pa_w = max((Wo - 1) * stride + k - pad - W, 0)
    xp = F.pad(x, (pad, pa_w, pad, pa_h))
    num = F.avg_pool2d(xp, k, stride)[:, :, :Ho, :Wo] * (k * k)
    if count_valid:
      ones = F.pad(torch.ones(1, 1, H, W), (pad, pa_w, pad, pa_h))
      den = F.avg_pool2d(ones, k, stride)[:, :, :Ho, :Wo] * (k * k)
      return num / den
    return num / (k * k)

  def asm_avgpool_fwd(self, x, y, N, H, W, Cn, k, stride, pad, Ho, Wo, cv, stream):
    out = self._avgpool(self._nchw(x, N, H, W, Cn), k, stride, pad, Ho, Wo, cv)
    T(y, (N, Ho, Wo, Cn), 'bf16').copy_(out.permute(0, 2, 3, 1))
    return 0

  def asm_avgpool_bwd(self, dy, dx, N, H, W, Cn, k, stride, pad, Ho, Wo, cv, addend, stream):
    if stride not in (1, 2):
      self._err = b'avgpool_bwd: stride not supported'
      return -1
    xt = torch.zeros(N, Cn, H, W, requires_grad=True)
    out = self._avgpool(xt, k, stride, pad, Ho, Wo, cv)
    g = T(dy, (N, Ho, Wo, Cn), 'bf16').float().permute(0, 3, 1, 2)
    (gx,) = torch.autograd.grad(out, xt, g)
    gx = gx.permute(0, 2, 3, 1)
    if addend:
      gx = gx + T(addend, (N, H, W, Cn), 'bf16').float()
    T(dx, (N, H, W, Cn), 'bf16').copy_(gx)
    return 0

  def asm_upsample2x_bwd(self, dy, dx, N, Hs, Ws, Cn, stream):
    g = T(dy, (N, Hs, 2, Ws, 2, Cn), 'bf16').float().sum(dim=(2, 4))
    T(dx, (N, Hs, Ws, Cn), 'bf16').copy_(g)
    return 0

  def asm_upsample2x_bwd_masked(self, dy, mask, dx, N, Hs, Ws, Cn, stream):
    g = T(dy, (N * Hs * 2 * Ws * 2, Cn), 'bf16').float()
    mk = T(mask, (N * Hs * 2 * Ws * 2, Cn // 8), 'u8').to(torch.int32)
    bits = ((mk[..., None] >> torch.arange(8, dtype=torch.int32)) & 1).view(-1, Cn).bool()
    g = torch.where(bits, g, torch.zeros_like(g)).view(N, Hs, 2, Ws, 2, Cn).sum(dim=(2, 4))
    T(dx, (N, Hs, Ws, Cn), 'bf16').copy_(g)
    return 0

  @staticmethod
  def _blur(x, k, stride):
    tri = {2: [1., 1.], 3: [1., 2., 1.], 4: [1., 3., 3., 1.], 5: [1., 4., 6., 4., 1.],
           6: [1., 5., 10., 10., 5., 1.], 7: [1., 6., 15., 20., 15., 6., 1.]}[k]
    a = torch.tensor(tri)
    f = a[:, None] * a[None, :]
    f = f / f.sum()
    p = (k - 1) // 2
    c = x.shape[1]
    xp = F.pad(x, (p, p, p, p), mode='reflect')
    return F.conv2d(xp, f.view(1, 1, k, k).repeat(c, 1, 1, 1), stride=stride, groups=c)

  def asm_blurpool_fwd(self, x, y, N, H, W, Cn, k, stride, stream):
    out = self._blur(self._nchw(x, N, H, W, Cn), k, stride)
    T(y, (N, out.shape[2], out.shape[3], Cn), 'bf16').copy_(out.permute(0, 2, 3, 1))
    return 0

  def asm_blurpool_bwd(self, dy, dx, N, H, W, Cn, k, stride, stream):
    xt = torch.zeros(N, Cn, H, W, requires_grad=True)
    out = self._blur(xt, k, stride)
    g = T(dy, (N, out.shape[2], out.shape[3], Cn), 'bf16').float().permute(0, 3, 1, 2)
    (gx,) = torch.autograd.grad(out, xt, g)
    T(dx, (N, H, W, Cn), 'bf16').copy_(gx.permute(0, 2, 3, 1))
    return 0

  def asm_gap_fwd(self, x, y, N, HW, Cn, stream):
    T(y, (N, Cn), 'bf16').copy_(T(x, (N, HW, Cn), 'bf16').float().mean(1))
    return 0

  def asm_gap_bwd(self, dy, dx, N, HW, Cn, stream):
    g = T(dy, (N, 1, Cn), 'bf16').float() / HW
    T(dx, (N, HW, Cn), 'bf16').copy_(g.expand(N, HW, Cn))
    return 0

  # ---- SK / SE -------------------------------------------------------------------------------------
  def asm_sk_gap(self, f, s, N, HW, F_, stream):
    ff = T(f, (N, HW, 2, F_), 'bf16').float()
    T(s, (N, F_), 'bf16').copy_(ff.sum(2).mean(1))
    return 0

  @staticmethod
  def _a0(att, N, F_):
    a = T(att, (N, 2, F_), 'f32')
    return torch.sigmoid(a[:, 0] - a[:, 1])

  def asm_sk_select_fwd(self, f, att, v, N, HW, F_, stream):
    ff = T(f, (N, HW, 2, F_), 'bf16').float()
    a0 = self._a0(att, N, F_)[:, None, :]
    T(v, (N, HW, F_), 'bf16').copy_(a0 * ff[:, :, 0] + (1 - a0) * ff[:, :, 1])
    return 0

  def asm_sk_select_bwd_att(self, f, dv, att, datt, N, HW, F_, stream):
    ff = T(f, (N, HW, 2, F_), 'bf16').float()
    g = T(dv, (N, HW, F_), 'bf16').float()
    a0 = self._a0(att, N, F_)
    t = ((ff[:, :, 0] - ff[:, :, 1]) * g).sum(1)
    d0 = a0 * (1 - a0) * t
    out = T(datt, (N, 2, F_), 'bf16')
    out[:, 0] = d0.to(torch.bfloat16)
    out[:, 1] = (-d0).to(torch.bfloat16)
    return 0

  def asm_sk_select_bwd_f(self, dv, att, ds, df, N, HW, F_, stream):
    g = T(dv, (N, HW, F_), 'bf16').float()
    a0 = self._a0(att, N, F_)[:, None, :]
    u = T(ds, (N, 1, F_), 'bf16').float() / HW
    out = T(df, (N, HW, 2, F_), 'bf16')
    out[:, :, 0] = (a0 * g + u).to(torch.bfloat16)
    out[:, :, 1] = ((1 - a0) * g + u).to(torch.bfloat16)
    return 0

  # ---- SK unit with the 3x3 conv's BN + ReLU applied on the fly ----
  def _f_from_y(self, y, scale, shift, N, HW, F_):
    yy = T(y, (N, HW, 2 * F_), 'bf16').float()
    f = torch.relu(yy * T(scale, (2 * F_,), 'f32') + T(shift, (2 * F_,), 'f32'))
    return f.to(torch.bfloat16).float().view(N, HW, 2, F_), yy

  def asm_sk_gap_bn(self, y, scale, shift, s, N, HW, F_, stream):
    f, _ = self._f_from_y(y, scale, shift, N, HW, F_)
    T(s, (N, F_), 'bf16').copy_((f[:, :, 0] + f[:, :, 1]).mean(1))
    return 0

  def asm_sk_select_bn_fwd(self, y, scale, shift, att, v, N, HW, F_, stream):
    f, _ = self._f_from_y(y, scale, shift, N, HW, F_)
    a0 = self._a0(att, N, F_)[:, None, :]
    T(v, (N, HW, F_), 'bf16').copy_(a0 * f[:, :, 0] + (1 - a0) * f[:, :, 1])
    return 0

  def asm_sk_select_bn_bwd_att(self, y, scale, shift, dv, att, datt, N, HW, F_, stream):
    f, _ = self._f_from_y(y, scale, shift, N, HW, F_)
    g = T(dv, (N, HW, F_), 'bf16').float()
    a0 = self._a0(att, N, F_)
    d0 = a0 * (1 - a0) * ((f[:, :, 0] - f[:, :, 1]) * g).sum(1)
    out = T(datt, (N, 2, F_), 'bf16')
    out[:, 0] = d0.to(torch.bfloat16)
    out[:, 1] = (-d0).to(torch.bfloat16)
    return 0

  def _sk_mask_y(self, y, scale, shift, N, HW, F_):
    yy = T(y, (N, HW, 2 * F_), 'bf16').float()
    m = ((yy * T(scale, (2 * F_,), 'f32') + T(shift, (2 * F_,), 'f32')) > 0).float()
    return m, yy

  def asm_sk_gap_bn_stats(self, y, scale, shift, mean, invstd, s, mst, N, HW, F_, stream):
    self.asm_sk_gap_bn(y, scale, shift, s, N, HW, F_, stream)
    m, yy = self._sk_mask_y(y, scale, shift, N, HW, F_)
    out = T(mst, (N, 2, 2 * F_), 'f32')
    out[:, 0] = m.sum(1)
    out[:, 1] = (m * yy).sum(1)
    return 0

  def asm_sk_select_bn_bwd_att_stats(self, y, scale, shift, mean, invstd, dv, att, datt, gst, N, HW, F_, stream):
    self.asm_sk_select_bn_bwd_att(y, scale, shift, dv, att, datt, N, HW, F_, stream)
    m, yy = self._sk_mask_y(y, scale, shift, N, HW, F_)
    g = T(dv, (N, HW, F_), 'bf16').float().repeat(1, 1, 2)
    out = T(gst, (N, 2, 2 * F_), 'f32')
    out[:, 0] = (m * g).sum(1)
    out[:, 1] = (m * g * yy).sum(1)
    return 0

  def asm_sk_bn_bwd_finalize(self, gst, mst, att, ds, N, HW, F_, gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, stream):
    C2 = 2 * F_
    a0 = self._a0(att, N, F_)
    ab = torch.cat([a0, 1 - a0], 1).double()                                   # [N, 2F]
    u = (T(ds, (N, F_), 'bf16').float() / HW).repeat(1, 2).double()
    gs, ms = T(gst, (N, 2, C2), 'f32').double(), T(mst, (N, 2, C2), 'f32').double()
    mu, isd, gm = T(mean, (C2,), 'f32').double(), T(invstd, (C2,), 'f32').double(), T(gamma, (C2,), 'f32').double()
    w = ab * gs[:, 0] + u * ms[:, 0]
    db = w.sum(0)
    dg = ((ab * gs[:, 1] + u * ms[:, 1] - mu * w).sum(0)) * isd
    M = N * HW
    T(dbeta, (C2,), 'f32').copy_(db.float())
    T(dgamma, (C2,), 'f32').copy_(dg.float())
    A = gm * isd
    B = -gm * isd * isd * dg / M
    T(cA, (C2,), 'f32').copy_(A.float())
    T(cB, (C2,), 'f32').copy_(B.float())
    T(cC, (C2,), 'f32').copy_((-gm * isd * db / M - B * mu).float())
    return 0

  def asm_sk_bn_bwd_blocks(self, N, HW, F_):
    return N

  def _sk_dz(self, dv, att, ds, y, scale, shift, N, HW, F_):
    f, yy = self._f_from_y(y, scale, shift, N, HW, F_)
    g = T(dv, (N, HW, 1, F_), 'bf16').float()
    a0 = self._a0(att, N, F_)[:, None, :]
    ab = torch.stack([a0, 1 - a0], 2)                      # [N,1,2,F]
    u = T(ds, (N, 1, 1, F_), 'bf16').float() / HW
    pre = yy * T(scale, (2 * F_,), 'f32') + T(shift, (2 * F_,), 'f32')
    df = (ab * g + u).reshape(N, HW, 2 * F_)
    return torch.where(pre > 0, df, torch.zeros_like(df)), yy

  def asm_sk_bn_bwd_reduce(self, dv, att, ds, y, scale, shift, mean, invstd, N, HW, F_, part, stream):
    dz, yy = self._sk_dz(dv, att, ds, y, scale, shift, N, HW, F_)
    xhat = (yy - T(mean, (2 * F_,), 'f32')) * T(invstd, (2 * F_,), 'f32')
    out = T(part, (N, 2, 2 * F_), 'f32')
    out[:, 0] = dz.sum(1)
    out[:, 1] = (dz * xhat).sum(1)
    return 0

  def asm_sk_bn_bwd_apply(self, dv, att, ds, y, scale, shift, cA, cB, cC, dy, N, HW, F_, stream):
    dz, yy = self._sk_dz(dv, att, ds, y, scale, shift, N, HW, F_)
    C2 = 2 * F_
    T(dy, (N, HW, C2), 'bf16').copy_(T(cA, (C2,), 'f32') * dz + T(cB, (C2,), 'f32') * yy + T(cC, (C2,), 'f32'))
    return 0

  def asm_se_scale_fwd(self, x, e, y, N, HW, Cn, stream):
    s = torch.sigmoid(T(e, (N, 1, Cn), 'f32'))
    T(y, (N, HW, Cn), 'bf16').copy_(T(x, (N, HW, Cn), 'bf16').float() * s)
    return 0

  def asm_se_scale_bwd_e(self, x, dy, e, de, N, HW, Cn, stream):
    s = torch.sigmoid(T(e, (N, Cn), 'f32'))
    t = (T(x, (N, HW, Cn), 'bf16').float() * T(dy, (N, HW, Cn), 'bf16').float()).sum(1)
    T(de, (N, Cn), 'bf16').copy_(t * s * (1 - s))
    return 0

  def asm_se_scale_bwd_x(self, dy, e, dsq, dx, N, HW, Cn, stream):
    s = torch.sigmoid(T(e, (N, 1, Cn), 'f32'))
    v = T(dy, (N, HW, Cn), 'bf16').float() * s + T(dsq, (N, 1, Cn), 'bf16').float() / HW
    T(dx, (N, HW, Cn), 'bf16').copy_(v)
    return 0

  # ---- element-wise --------------------------------------------------------------------------------
  def asm_relu_fwd(self, x, y, n, stream):
    T(y, (n,), 'bf16').copy_(F.relu(T(x, (n,), 'bf16').float()))
    return 0

  def asm_relu_bwd(self, dy, y, dx, n, stream):
    T(dx, (n,), 'bf16').copy_(T(dy, (n,), 'bf16').float() * (T(y, (n,), 'bf16').float() > 0))
    return 0

  def asm_add_bf16(self, a, b, out, n, stream):
    v = T(a, (n,), 'bf16').float() + T(b, (n,), 'bf16').float()
    T(out, (n,), 'bf16').copy_(v)
    return 0

  def asm_bias_add_f32(self, y, bias, M, Cn, ldy, stream):
    T(y, (M, ldy), 'f32')[:, :Cn] += T(bias, (Cn,), 'f32')
    return 0

  def asm_bias_grad_bf16(self, dz, M, Cn, ld, dbias, stream):
    T(dbias, (Cn,), 'f32').copy_(T(dz, (M, ld), 'bf16').float()[:, :Cn].sum(0))
    return 0

  def asm_cast_f32_to_bf16(self, x, y, n, stream):
    if n:
      T(y, (n,), 'bf16').copy_(T(x, (n,), 'f32'))
    return 0

  def asm_cast_bf16_to_f32(self, x, y, n, stream):
    if n:
      T(y, (n,), 'f32').copy_(T(x, (n,), 'bf16'))
    return 0

  # ---- loss ----------------------------------------------------------------------------------------
  def asm_softmax_ce(self, logits, ld, targets, teacher, B, Cn, eps, T_, loss_scale, loss_rows, dlogits, ld_out,
                     stream):
    z = T(logits, (B, ld), 'f32')[:, :Cn].clone().requires_grad_(True)
    y = T(targets, (B, Cn), 'f32') * (1 - eps) + eps / Cn
    rows = -(y * torch.log_softmax(z, 1)).sum(1)
    if teacher:
      t = T(teacher, (B, Cn), 'f32')
      rows = rows + T_ * T_ * (-(t * torch.log_softmax(z / T_, 1)).sum(1))
    T(loss_rows, (B,), 'f32').copy_(rows.detach())
    if dlogits:
      (g,) = torch.autograd.grad(rows.sum() * (loss_scale / B), z)
      out = T(dlogits, (B, ld_out), 'bf16')
      out.zero_()
      out[:, :Cn] = g.to(torch.bfloat16)
    return 0

  def asm_onehot(self, labels, out, B, Cn, stream):
    T(out, (B, Cn), 'f32').copy_(F.one_hot(T(labels, (B,), 'i32').long(), Cn).float())
    return 0

  def asm_softmax_rows(self, x, y, B, Cn, inv_temp, stream):
    T(y, (B, Cn), 'f32').copy_(torch.softmax(T(x, (B, Cn), 'f32') * inv_temp, 1))
    return 0

  def asm_mean_f32(self, x, n, out, stream):
    T(out, (1,), 'f32').copy_(T(x, (n,), 'f32').mean().view(1))
    return 0

  # ---- input ---------------------------------------------------------------------------------------
  @staticmethod
  def _mix(v, Bin, mixup_type, lam1, lam2):
    if mixup_type == 0:
      return v
    shape = [-1] + [1] * (v.dim() - 1)
    half = Bin // 2
    l1 = T(lam1, (half,), 'f32').view(shape)
    first = l1 * v[:half] + (1 - l1) * v[half:]
    if mixup_type == 1:
      return first
    l2 = T(lam2, (half,), 'f32').view(shape)
    second = l2 * v[:half] + (1 - l2) * torch.flip(v[half:], [0])
    return torch.cat([first, second], 0)

  def asm_mixup_meansub(self, images, is_u8, Bin, H, W, mixup_type, lam1, lam2, out, stream):
    img = T(images, (Bin, H, W, 3), 'u8' if is_u8 else 'f32').float()
    img = img - torch.tensor([123.68, 116.78, 103.94])
    mixed = self._mix(img, Bin, mixup_type, lam1, lam2)
    Bout = mixed.shape[0]
    dst = T(out, (Bout, H + 6, W + 6, 4), 'bf16')
    dst.zero_()
    dst[:, 3:3 + H, 3:3 + W, :3] = mixed.to(torch.bfloat16)
    return 0

  def asm_mixup_labels(self, y, Bin, Cn, mixup_type, lam1, lam2, out, stream):
    mixed = self._mix(T(y, (Bin, Cn), 'f32'), Bin, mixup_type, lam1, lam2)
    T(out, tuple(mixed.shape), 'f32').copy_(mixed)
    return 0

  # ---- optimiser -----------------------------------------------------------------------------------
  def asm_sgd_momentum(self, w, accum, grad, wb, n, lr, momentum, wd, gs, stream):
    if n == 0:
      return 0
    tw, ta, tg = T(w, (n,), 'f32'), T(accum, (n,), 'f32'), T(grad, (n,), 'f32')
    gg = tg * gs + wd * tw
    ta.copy_(momentum * ta + gg)
    tw.copy_(tw - lr * ta)
    if wb:
      T(wb, (n,), 'bf16').copy_(tw)
    return 0

  # ---- sigmoid loss / GeM / DropBlock / eval metrics ---------------------------------------------------
  def asm_sigmoid_ce(self, logits, ld, targets, B, Cn, loss_scale, rows, out, dlogits, ld_out, stream):
    z = T(logits, (B, ld), 'f32')[:, :Cn].clone().requires_grad_(True)
    y = T(targets, (B, Cn), 'f32')
    ce = F.binary_cross_entropy_with_logits(z, y, reduction='sum')
    ys = y.sum()
    loss = ce / ys
    o = T(out, (2,), 'f32')
    o[0] = loss.detach()
    o[1] = ys
    if dlogits:
      (g,) = torch.autograd.grad(loss * loss_scale, z)
      d = T(dlogits, (B, ld_out), 'bf16')
      d.zero_()
      d[:, :Cn] = g.to(torch.bfloat16)
    return 0

  def asm_gem_fwd(self, x, y, ssum, N, HW, Cn, p, stream):
    v = T(x, (N, HW, Cn), 'bf16').float().clamp(1e-6, 1e12)
    s = (v ** p).sum(1).clamp(min=1e-6)
    T(ssum, (N, Cn), 'f32').copy_(s)
    T(y, (N, Cn), 'bf16').copy_((HW ** (-1.0 / p)) * s ** (1.0 / p))
    return 0

  def asm_gem_bwd(self, x, dy, ssum, dx, N, HW, Cn, p, stream):
    v = T(x, (N, HW, Cn), 'bf16').float().requires_grad_(True)
    out = (HW ** (-1.0 / p)) * ((v.clamp(1e-6, 1e12) ** p).sum(1).clamp(min=1e-6)) ** (1.0 / p)
    (g,) = torch.autograd.grad(out, v, T(dy, (N, Cn), 'bf16').float())
    T(dx, (N, HW, Cn), 'bf16').copy_(g)
    return 0

  def asm_dropblock_mask(self, uniform, gamma, H, W, Cn, bs, keep, scale, stream):
    u = T(uniform, (H - bs + 1, W - bs + 1, Cn), 'f32').permute(2, 0, 1)[None]
    br = (bs - 1) // 2
    tl = (bs - 1) - br
    m = F.relu(torch.sign(gamma - u))
    m = F.pad(m, (tl, br, tl, br))
    pb = (bs - 1) // 2
    m = F.max_pool2d(F.pad(m, (pb, bs - 1 - pb, pb, bs - 1 - pb), value=float('-inf')), bs, 1)
    k = (1 - m)[0].permute(1, 2, 0).contiguous()
    T(keep, (H, W, Cn), 'f32').copy_(k)
    T(scale, (1,), 'f32').copy_((k.numel() / (k.sum() + 1e-8)).view(1))
    return 0

  def asm_dropblock_mask_dev(self, uniform, gamma_dev, H, W, Cn, bs, keep, scale, stream):
    return self.asm_dropblock_mask(uniform, float(T(gamma_dev, (1,), 'f32')[0]), H, W, Cn, bs, keep, scale, stream)

  def asm_allreduce_bucket(self, buf, count, dtype, comm, comm_stream, producer_stream):
    self._err = b'allreduce_bucket: the CPU test double has no RCCL'
    return -2

  def asm_memcpy_async(self, dst, src, nbytes, stream):
    T(dst, (nbytes,), 'u8').copy_(T(src, (nbytes,), 'u8'))
    return 0

  def asm_dropblock_apply(self, x, keep, scale, relu_mask_from, relu, y, N, HWC, stream):
    v = T(x, (N, HWC), 'bf16').float() * T(keep, (HWC,), 'f32') * T(scale, (1,), 'f32')
    if relu_mask_from:
      v = v * (T(relu_mask_from, (N, HWC), 'bf16').float() > 0)
    elif relu:
      v = F.relu(v)
    T(y, (N, HWC), 'bf16').copy_(v)
    return 0

  def asm_eval_rows(self, logits, ld, labels, B, Cn, pred, conf, top1, top5, stream):
    z = T(logits, (B, ld), 'f32')[:, :Cn]
    lab = T(labels, (B,), 'i32').long()
    p = z.argmax(1)
    T(pred, (B,), 'i32').copy_(p.to(torch.int32))
    T(conf, (B,), 'f32').copy_(torch.softmax(z, 1).max(1).values)
    T(top1, (B,), 'f32').copy_((p == lab).float())
    zl = z.gather(1, lab[:, None])
    T(top5, (B,), 'f32').copy_(((z > zl).sum(1) < 5).float())
    return 0

  def asm_eval_accumulate(self, conf, top1, top5, B, state, stream):
    c, t1, t5 = T(conf, (B,), 'f32'), T(top1, (B,), 'f32'), T(top5, (B,), 'f32')
    st = T(state, (33,), 'f32')
    st[0] += t1.sum()
    st[1] += t5.sum()
    st[2] += B
    for b in range(10):
      lo = -1e-7 if b == 0 else b / 10.0
      hi = 1 + 1e-7 if b == 9 else (b + 1) / 10.0
      sel = (c > lo) & (c <= hi)
      st[3 + b] += t1[sel].sum()
      st[13 + b] += c[sel].sum()
      st[23 + b] += sel.sum()
    return 0

  def asm_resize_crop_flip(self, src, src_bytes, descs, N, out_h, out_w, subtract_mean, out, stream):
    from assembled_cnn_amd.lib import ImageDesc
    if N < 0 or out_h <= 0 or out_w <= 0 or src_bytes < 0:
      self._err = b'resize_crop_flip: bad sizes'
      return -1
    if N == 0:
      return 0
    buf = T(src, (src_bytes,), 'u8')
    o = T(out, (N, out_h, out_w, 3), 'f32')
    table = (ImageDesc * N).from_address(descs)
    means = torch.tensor([123.68, 116.78, 103.94])
    for n in range(N):
      d = table[n]
      ok = (d.src_offset >= 0 and d.src_offset + d.Hs * d.Ws * 3 <= src_bytes and d.crop_h > 0 and d.crop_w > 0 and
            d.crop_y >= 0 and d.crop_x >= 0 and d.crop_y + d.crop_h <= d.Hs and d.crop_x + d.crop_w <= d.Ws and
            d.resize_h > 0 and d.resize_w > 0 and d.out_y >= 0 and d.out_x >= 0 and
            d.out_y + out_h <= d.resize_h and d.out_x + out_w <= d.resize_w)
      if not ok:
        o[n].zero_()
        continue
      img = buf[d.src_offset:d.src_offset + d.Hs * d.Ws * 3].view(d.Hs, d.Ws, 3)
      win = img[d.crop_y:d.crop_y + d.crop_h, d.crop_x:d.crop_x + d.crop_w].float()
      if d.flip:
        win = win.flip(1)
      hs = torch.tensor(float(d.crop_h)) / torch.tensor(float(d.resize_h))
      ws = torch.tensor(float(d.crop_w)) / torch.tensor(float(d.resize_w))
      sy = (torch.arange(out_h) + d.out_y).float() * hs
      sx = (torch.arange(out_w) + d.out_x).float() * ws
      ly, lx = sy.long(), sx.long()
      uy, ux = (ly + 1).clamp(max=d.crop_h - 1), (lx + 1).clamp(max=d.crop_w - 1)
      fy, fx = (sy - ly.float())[:, None, None], (sx - lx.float())[None, :, None]
      top = win[ly][:, lx] + (win[ly][:, ux] - win[ly][:, lx]) * fx
      bot = win[uy][:, lx] + (win[uy][:, ux] - win[uy][:, lx]) * fx
      v = top + (bot - top) * fy
      o[n].copy_(v - means if subtract_mean else v)
    return 0
