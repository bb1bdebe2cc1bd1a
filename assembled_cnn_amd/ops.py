"""Tensor-level wrappers over the C ABI (include/asm_hip.h).

PyTorch is used only for device memory (torch.empty), streams and tensor handles; every arithmetic
op below is a HIP kernel in libasm_hip.so.  There is no CPU fallback: tensors must live on the GPU.
(The test-suite substitutes a CPU double of the C ABI via ``set_library`` to exercise the host logic
in this package without a GPU; the double lives under tests/ and is never used by the product.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import lib as _lib
from .lib import ConvDesc, check

_L = None          # the bound library (or the test double)
_IS_DOUBLE = False


def set_library(obj, is_double: bool = False):
  """Install the C-ABI provider (tests only)."""
  global _L, _IS_DOUBLE
  _L = obj
  _IS_DOUBLE = is_double


def L():
  global _L
  if _L is None:
    _L = _lib.load()
  return _L


def abi_calls() -> int:
  """number of C-ABI calls made through this module so far"""
  return _lib.CALLS[0]


_KNOBS = {}


def knob(name: str, default: str) -> str:
  """An ASM_* host-side switch.  Read from the environment ONCE (a training step asks ~1500 times: os.environ.get was
  2 ms of host time per step) and again after refresh_tuning()."""
  v = _KNOBS.get(name)
  if v is None:
    v = _KNOBS[name] = os.environ.get(name, default)
  return v


def refresh_tuning():
  """Re-read the ASM_* variables of THIS process: the host-side switches (knob) and the kernel-selection ones, which are
  handed to the library (asm_set_tuning).  The library reads no environment itself; lib.load() does this once, tests /
  A/B tools call it after changing a variable."""
  _KNOBS.clear()
  if _IS_DOUBLE:
    return None
  return _lib.apply_env_tuning(L())


def _ptr(t: Optional[torch.Tensor]) -> int:
  if t is None:
    return 0
  if not _IS_DOUBLE and not t.is_cuda:
    raise _lib.AsmError('assembled_cnn_amd ops need GPU tensors (no CPU fallback)')
  if not t.is_contiguous():
    raise ValueError('tensor must be contiguous')
  return t.data_ptr()


# torch.cuda.current_stream() builds a Stream object through four Python layers (~8 us; ~1200 calls per training step);
# the raw handle of the current stream is one C call away
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


import threading  # noqa: E402

# per host THREAD: an input-pipeline or evaluation thread that calls ops while the training thread has a weight gradient
# redirected to a side stream keeps launching on its own current stream
_TLS = threading.local()


def launch_on(stream: Optional["torch.cuda.Stream"]):
  """Every following C-ABI call OF THIS THREAD goes to ``stream`` instead of torch's current stream (None = back to the
  current stream).  For launch-only regions -- nothing inside may allocate through torch, whose caching allocator keys
  blocks by ITS current stream: the weight-gradient side stream uses it instead of a ``with torch.cuda.stream(...)`` block
  (~15 us of host time per use, 117 uses per step)."""
  _TLS.stream = None if stream is None else stream.cuda_stream


def _stream() -> int:
  if _IS_DOUBLE:
    return 0
  ov = getattr(_TLS, 'stream', None)
  if ov is not None:
    return ov
  if _RAW_STREAM is not None and _CUR_DEVICE is not None:
    return _RAW_STREAM(_CUR_DEVICE())
  return torch.cuda.current_stream().cuda_stream


def stream_join(dst: "torch.cuda.Stream", src: "torch.cuda.Stream"):
  """``dst`` waits for everything enqueued on ``src`` so far (== dst.wait_stream(src)), through the library: every
  cross-stream edge of a training step goes through here so that a launch tape that is being recorded sees it
  (csrc/tape.hip)."""
  if dst is src:
    return
  if _IS_DOUBLE:            # the CPU test double has no streams; its tests pass stand-ins that log who waited for whom
    dst.wait_stream(src)
    return
  check(L().asm_stream_join(dst.cuda_stream, src.cuda_stream), 'stream_join')


def tape_begin() -> int:
  rc = L().asm_tape_begin()
  if rc <= 0:
    check(rc, 'tape_begin')
  return rc


def tape_mark() -> int:
  rc = L().asm_tape_mark()
  if rc <= 0:
    check(rc, 'tape_mark')
  return rc


def tape_end() -> int:
  rc = L().asm_tape_end()
  if rc <= 0:
    check(rc, 'tape_end')
  return rc


def tape_info(tape: int) -> dict:
  import ctypes
  info = (ctypes.c_int64 * 6)()
  check(L().asm_tape_info(tape, ctypes.byref(info)), 'tape_info')
  return dict(zip(('nodes', 'launches', 'joins', 'fills', 'segments', 'arg_bytes'), [int(v) for v in info]))


def tape_replay(tape: int, segment: int = -1):
  check(L().asm_tape_replay(tape, segment), 'tape_replay')


def tape_free(tape: int):
  check(L().asm_tape_free(tape), 'tape_free')


def empty(shape, dtype, like: torch.Tensor) -> torch.Tensor:
  return torch.empty(shape, dtype=dtype, device=like.device)


BF16 = torch.bfloat16
F32 = torch.float32


# ---------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------
def out_size(in_size: int, k: int, stride: int) -> int:
  """conv2d_fixed_padding output size (nets/model_helper.py:67-78): SAME for stride 1,
  fixed_padding(k) + VALID for stride > 1 -> (in + k - 1 - k) // s + 1."""
  if stride == 1:
    return in_size
  return (in_size - 1) // stride + 1


def make_conv_desc(N, H, W, Cin, K, R, S, stride, pad=None, Ho=None, Wo=None, ldy=0, out_f32=False,
                   img_pitch=0, row_pitch=0, pix_pitch=0) -> ConvDesc:
  d = ConvDesc()
  d.N, d.H, d.W, d.C = N, H, W, Cin
  d.K, d.R, d.S = K, R, S
  d.stride = stride
  d.pad = (R - 1) // 2 if pad is None else pad
  d.Ho = out_size(H, R, stride) if Ho is None else Ho
  d.Wo = out_size(W, S, stride) if Wo is None else Wo
  d.x_img_pitch, d.x_row_pitch, d.x_pix_pitch = img_pitch, row_pitch, pix_pitch
  d.ldy = ldy
  d.out_f32 = 1 if out_f32 else 0
  return d


class ConvTimer(object):
  """HIP-event timing of kernel launches on the launch stream (bench.py's roofline leg).
  ``only`` restricts convolution timing to one (kind, shape-key) class; otherwise every conv launch is timed.
  With ``classes`` the batch-norm family is timed too, as one class with its algorithmic bytes per call."""

  def __init__(self, only=None, classes=False):
    self.only = only
    self.classes = classes
    self.events = {}
    self.class_events = {}

  @staticmethod
  def key(kind, d):
    return (kind, d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride)

  def start(self, kind, d):
    k = self.key(kind, d)
    if self.only is not None and k != self.only:
      return None
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    self.events.setdefault(k, []).append((e0, e1))
    return e1

  def start_class(self, cls, work):
    """-> the end event (record it after the launches); ``work`` = algorithmic bytes (or FLOPs) of the call"""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    self.class_events.setdefault(cls, []).append((e0, e1, float(work)))
    return e1

  def summary(self):
    """key -> (launches, total ms); call after a device synchronize."""
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.events.items()}

  def class_summary(self):
    """class -> (calls, total ms, total work)"""
    return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v), sum(w for _, _, w in v))
            for k, v in self.class_events.items()}


_TIMER: Optional[ConvTimer] = None


def set_conv_timer(t: Optional[ConvTimer]):
  global _TIMER
  _TIMER = t


def timer_on() -> bool:
  return _TIMER is not None


def _bn_ev(nbytes):
  return _TIMER.start_class('bn', nbytes) if (_TIMER is not None and _TIMER.classes) else None


def dense_small_on() -> bool:
  """ASM_DENSE_SMALL=0 keeps the [N,1,1,C] layers on the implicit-GEMM convolution (A/B runs, tests); cached until refresh_tuning()."""
  return knob('ASM_DENSE_SMALL', '1') != '0'


def _is_dense(d: ConvDesc) -> bool:
  return (d.H == 1 and d.W == 1 and d.Ho == 1 and d.Wo == 1 and d.R == 1 and d.S == 1 and d.x_img_pitch in (0, d.C)
          and d.x_row_pitch in (0, d.C) and d.x_pix_pitch in (0, d.C) and d.C % 8 == 0)


def conv_fprop(d: ConvDesc, x: torch.Tensor, w: torch.Tensor, want_stats: bool = False
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
  """y [N,Ho,Wo,ldy] (bf16 or f32) and, if want_stats, the BN partials [blocks,2,K] (f32)."""
  ldy = d.ldy if d.ldy else d.K
  y = empty((d.N, d.Ho, d.Wo, ldy), F32 if d.out_f32 else BF16, x)
  stats = None
  if want_stats:
    stats = empty((L().asm_conv2d_stats_blocks(C.byref(d)), 2, d.K), F32, x)
  ev = _TIMER.start('fprop', d) if _TIMER is not None else None
  if not want_stats and _is_dense(d) and d.C % 16 == 0 and dense_small_on():
    # [N,1,1,C] squeeze / excite / classifier layer: row-major product, no implicit-GEMM machinery (csrc/dense_small.hip)
    check(L().asm_dense_small(_ptr(x), d.C, _ptr(w), d.C, d.N, d.K, d.C, _ptr(y), ldy, 1 if d.out_f32 else 0, None,
                              _stream()), 'dense_small')
  else:
    check(L().asm_conv2d_fprop(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(stats), _stream()), 'conv2d_fprop')
  if ev is not None:
    ev.record()
  return y, stats


def conv_fprop_bn(d: ConvDesc, x: torch.Tensor, w: torch.Tensor, scale, shift, residual=None, relu=False):
  """Inference: conv + folded batch norm [+ residual] [+ ReLU] in one launch -> y bf16 [N, Ho, Wo, K]."""
  y = empty((d.N, d.Ho, d.Wo, d.K), BF16, x)
  ev = _TIMER.start('fprop', d) if _TIMER is not None else None
  check(L().asm_conv2d_fprop_bn(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(scale), _ptr(shift), _ptr(residual),
                                1 if relu else 0, _stream()), 'conv2d_fprop_bn')
  if ev is not None:
    ev.record()
  return y


def dgrad_pool_ok(d: ConvDesc) -> bool:
  """can asm_conv2d_dgrad_pooled take this layer (1x1, stride 1, on the igemm2 path)?  ASM_POOL_FUSE=0: never"""
  return (knob('ASM_POOL_FUSE', '1') != '0' and d.R == 1 and d.S == 1 and d.stride == 1 and d.pad == 0
          and d.C % 8 == 0 and d.K % 32 == 0 and not _is_dense(d) and knob('ASM_IGEMM_MODE', '0') in ('', '0'))


def dgrad_s2_ok(d: ConvDesc) -> bool:
  """does the one-launch 3x3 / stride-2 input gradient (csrc/conv_dgrad_s2.hip) take this layer?  It adds a MASKED fan-in
  addend in its copy-out, so the caller need not materialise the masked gradient first.  ASM_DGRAD_PARITY < 2: never"""
  if not _IS_DOUBLE:        # the library decides from asm_tuning (asm_dgrad_s2_try): ask it, not only the environment
    t = _lib.Tuning()
    L().asm_get_tuning(C.byref(t))
    if t.dgrad_parity < 2 or t.igemm_mode:
      return False
  return (knob('ASM_DGRAD_PARITY', '2') == '2' and knob('ASM_IGEMM_MODE', '0') in ('', '0') and d.R == 3 and d.S == 3
          and d.stride == 2 and d.pad == 1 and d.C == 64 and d.K == 64 and d.H == 2 * d.Ho and d.W == 2 * d.Wo
          and d.Ho % 8 == 0 and d.Wo % 8 == 0)


def dgrad_bnred_ok(d: ConvDesc) -> bool:
  """should this layer's input gradient reduce the batch-norm backward sums of its input (asm_conv2d_dgrad_bnred)?  Whatever the
  library covers (asm_conv2d_dgrad_bnred_blocks > 0): the 1x1 stride-1 layers and the 3x3 stride-1 layers of its igemm8 / igemm3
  kernels.  ASM_BN_RED=0: never; ASM_BN_RED=1x1: the 1x1 layers only (round 6 A/B)"""
  mode = knob('ASM_BN_RED', '1')
  if mode == '0' or _is_dense(d) or d.stride != 1 or d.C % 8:
    return False
  if mode == '1x1' and d.R != 1:
    return False
  return L().asm_conv2d_dgrad_bnred_blocks(C.byref(d)) > 0


def conv_dgrad_bnred(d: ConvDesc, dy: torch.Tensor, wt: torch.Tensor, addend, addend_mask, bn_y: torch.Tensor, bn_mask):
  """conv_dgrad + the reduce pass of the batch-norm backward of the layer that produced this convolution's input
  (asm_conv2d_dgrad_bnred) -> (dx, partial [blocks][2][C] of (sum dz, sum dz * y))"""
  dx = empty((d.N, d.H, d.W, d.C), BF16, dy)
  blocks = L().asm_conv2d_dgrad_bnred_blocks(C.byref(d))
  part = empty((blocks, 2, d.C), F32, dy)
  ev = _TIMER.start('dgrad', d) if _TIMER is not None else None
  check(L().asm_conv2d_dgrad_bnred(C.byref(d), _ptr(dy), _ptr(wt), _ptr(addend), _ptr(addend_mask), _ptr(bn_y), _ptr(bn_mask),
                                   _ptr(part), _ptr(dx), _stream()), 'conv2d_dgrad_bnred')
  if ev is not None:
    ev.record()
  return dx, part


def conv_dgrad(d: ConvDesc, dy: torch.Tensor, wt: torch.Tensor, addend: Optional[torch.Tensor] = None,
               addend_mask: Optional[torch.Tensor] = None, pool=None) -> torch.Tensor:
  """dx = conv_transpose(dy, w) [+ addend [where addend_mask]] [+ avgpool_bwd(pool)];
  pool = (pooled gradient [N,Hp,Wp,C], k, stride, pad, count_valid)"""
  dx = empty((d.N, d.H, d.W, d.C), BF16, dy)
  ev = None
  if _TIMER is not None:     # tools/insitu_sweep.py keeps the input gradients with a fan-in addend apart (another kernel variant)
    ev = _TIMER.start('dgrad+add' if (addend is not None and getattr(_TIMER, 'split_addend', False)) else 'dgrad', d)
  if pool is not None:
    pdy, pk, pst, ppad, pcv = pool
    check(L().asm_conv2d_dgrad_pooled(C.byref(d), _ptr(dy), _ptr(wt), _ptr(addend), _ptr(addend_mask), _ptr(pdy), pk, pst,
                                      ppad, pdy.shape[1], pdy.shape[2], 1 if pcv else 0, _ptr(dx), _stream()),
          'conv2d_dgrad_pooled')
    if ev is not None:
      ev.record()
    return dx
  if addend_mask is None and _is_dense(d) and d.K % 16 == 0 and dense_small_on():
    # dy [N][K] (row stride K: a padded Cout arrives as K = kpad), wt = CRSK copy [C][K]
    check(L().asm_dense_small(_ptr(dy), d.K, _ptr(wt), d.K, d.N, d.C, d.K, _ptr(dx), d.C, 0, _ptr(addend), _stream()),
          'dense_small')
  elif addend_mask is not None:
    check(L().asm_conv2d_dgrad_masked(C.byref(d), _ptr(dy), _ptr(wt), _ptr(addend), _ptr(addend_mask), _ptr(dx),
                                      _stream()), 'conv2d_dgrad_masked')
  else:
    check(L().asm_conv2d_dgrad(C.byref(d), _ptr(dy), _ptr(wt), _ptr(addend), _ptr(dx), _stream()), 'conv2d_dgrad')
  if ev is not None:
    ev.record()
  return dx


_ws_cache = {}
_ws_retired = []


def _workspace(nbytes: int, like: torch.Tensor) -> Optional[torch.Tensor]:
  """One grow-only scratch buffer per (device, stream): launches on one stream run in order, launches on two
  streams (the opt-in weight-gradient side stream) must not share the split-K slab."""
  if nbytes == 0:
    return None
  key = (str(like.device), _stream())
  buf = _ws_cache.get(key)
  if buf is None or buf.numel() < nbytes:
    if buf is not None:
      # Outgrown buffers are kept, not freed: under launch_on() the caching allocator does not know which stream still
      # reads the old one (it would recycle it for the stream that allocated it).  A handful per process, at start-up.
      _ws_retired.append(buf)
    buf = torch.empty((nbytes,), dtype=torch.uint8, device=like.device)
    _ws_cache[key] = buf
  return buf


def conv_wgrad(d: ConvDesc, x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor):
  """dw (f32, [K,R,S,C] contiguous view, overwritten)."""
  need = L().asm_conv2d_wgrad_workspace_bytes(C.byref(d))
  ws = _workspace(need, x)
  ev = _TIMER.start('wgrad', d) if _TIMER is not None else None
  if _is_dense(d) and d.N <= 1024 and dense_small_on():     # [N,1,1,C] layer: dw = dy^T . x over a few hundred rows
    check(L().asm_dense_small_wgrad(_ptr(x), d.C, _ptr(dy), d.ldy if d.ldy else d.K, d.N, d.C, d.K, _ptr(dw), d.C,
                                    _stream()), 'dense_small_wgrad')
    if ev is not None:
      ev.record()
    return
  check(L().asm_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), need, _stream()), 'conv2d_wgrad')
  if ev is not None:
    ev.record()


def filter_transpose(w: torch.Tensor, wt: torch.Tensor, K, R, S, Cin, ldk=0):
  check(L().asm_filter_transpose(_ptr(w), _ptr(wt), K, R, S, Cin, ldk, _stream()), 'filter_transpose')


def filter_transpose_batched(w_arena, wt_arena, table, nlayers, total):
  check(L().asm_filter_transpose_batched(_ptr(w_arena), _ptr(wt_arena), _ptr(table), nlayers, total, _stream()),
        'filter_transpose_batched')


def filter_transpose_tiled(w_arena, wt_arena, table, nlayers, total_tiles):
  check(L().asm_filter_transpose_tiled(_ptr(w_arena), _ptr(wt_arena), _ptr(table), nlayers, total_tiles, _stream()),
        'filter_transpose_tiled')


def stem_pack_filter(w32: torch.Tensor, wp: torch.Tensor, K: int, ksize: int):
  check(L().asm_stem_pack_filter(_ptr(w32), _ptr(wp), K, ksize, _stream()), 'stem_pack_filter')


def stem_unpack_grad(dwp: torch.Tensor, dw: torch.Tensor, K: int, ksize: int):
  check(L().asm_stem_unpack_grad(_ptr(dwp), _ptr(dw), K, ksize, _stream()), 'stem_unpack_grad')


def stem_pad_input(x: torch.Tensor) -> torch.Tensor:
  """[N,H,W,3] f32/bf16 -> [N,H+6,W+6,4] bf16 zero halo."""
  N, H, W, c = x.shape
  if c != 3:
    raise ValueError('stem input must have 3 channels')
  if x.dtype not in (F32, BF16):
    raise ValueError('stem input dtype must be float32 or bfloat16')
  xp = empty((N, H + 6, W + 6, 4), BF16, x)
  check(L().asm_stem_pad_input(_ptr(x), 1 if x.dtype == F32 else 0, _ptr(xp), N, H, W, _stream()), 'stem_pad_input')
  return xp


# ---------------------------------------------------------------------------------------------------
# batch norm
# ---------------------------------------------------------------------------------------------------
def bn_stats(x2d: torch.Tensor, M: int, Cn: int) -> torch.Tensor:
  blocks = L().asm_bn_stats_blocks(M, Cn)
  if blocks <= 0:
    raise ValueError('bn_stats: bad shape')
  part = empty((blocks, 2, Cn), F32, x2d)
  check(L().asm_bn_stats(_ptr(x2d), M, Cn, _ptr(part), _stream()), 'bn_stats')
  return part


def _compact(part: torch.Tensor, Cn: int) -> torch.Tensor:
  """Thousands of partial rows (conv epilogue at high resolution) -> <= 32 rows, fully parallel."""
  blocks = part.shape[0]
  if blocks <= 1024:
    return part
  per = -(-blocks // 32)
  groups = -(-blocks // per)
  out = empty((groups, 2, Cn), F32, part)
  check(L().asm_bn_partials_compact(_ptr(part), blocks, Cn, _ptr(out), groups, _stream()), 'bn_partials_compact')
  return out


def bn_finalize(part, M, Cn, gamma, beta, eps, momentum, mm, mv):
  """-> mean, invstd, scale, shift (each [C] f32); updates moving stats in place when given."""
  ev = _bn_ev(part.numel() * 4.0)
  part = _compact(part, Cn)
  co = empty((4, Cn), F32, part)
  check(L().asm_bn_finalize(_ptr(part), part.shape[0], M, Cn, _ptr(gamma), _ptr(beta), eps, momentum,
                            _ptr(mm), _ptr(mv), _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _ptr(co[3]),
                            _stream()), 'bn_finalize')
  if ev is not None:
    ev.record()
  return co[0], co[1], co[2], co[3]


def bn_infer_coeffs(Cn, gamma, beta, mm, mv, eps):
  co = empty((2, Cn), F32, gamma)
  check(L().asm_bn_infer_coeffs(Cn, _ptr(gamma), _ptr(beta), _ptr(mm), _ptr(mv), eps, _ptr(co[0]), _ptr(co[1]),
                                _stream()), 'bn_infer_coeffs')
  return co[0], co[1]


def bn_apply(x, M, Cn, scale, shift, residual=None, res_mode=0, relu=False, H=0, W=0, want_mask=False):
  """-> y, or (y, packed ReLU mask [M, C/8] uint8) with want_mask."""
  y = torch.empty_like(x)
  mask = empty((M, Cn // 8), torch.uint8, x) if (want_mask and relu) else None
  ev = _bn_ev(M * Cn * (4.0 + (2.0 if res_mode == 1 else 0.5 if res_mode == 2 else 0.0) + (0.125 if mask is not None else 0.0)))
  check(L().asm_bn_apply(_ptr(x), _ptr(y), M, Cn, _ptr(scale), _ptr(shift), _ptr(residual), res_mode,
                         1 if relu else 0, H, W, _ptr(mask), _stream()), 'bn_apply')
  if ev is not None:
    ev.record()
  return (y, mask) if want_mask else y


def bn_bwd(dy, x, yout, relu, M, Cn, gamma, mean, invstd, dgamma, dbeta, want_dz, raw_part=None):
  """-> dx, dz (dz None unless want_dz).  dgamma/dbeta: f32 [C] views, overwritten.
  ``yout`` is the bf16 forward output or (uint8) the packed ReLU mask from bn_apply(want_mask=True).
  ``raw_part``: the (sum dz, sum dz * y) partials the input gradient that wrote ``dy`` already reduced in its epilogue
  (conv_dgrad_bnred): the reduce pass over (dy, x) is skipped."""
  rk = 0 if not relu else (2 if yout.dtype == torch.uint8 else 1)
  red_bytes = 0.0 if raw_part is not None else (4.0 + (0.125 if rk == 2 else 2.0 if rk == 1 else 0.0))
  ev = _bn_ev(M * Cn * (red_bytes + 4.0 + 2.0 + (2.0 if want_dz else 0.0) + (0.125 if rk == 2 else 2.0 if rk == 1 else 0.0)))
  co = empty((3, Cn), F32, dy)
  if raw_part is not None:
    part = _compact(raw_part, Cn)
    check(L().asm_bn_bwd_finalize_raw(_ptr(part), part.shape[0], M, Cn, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                                      _ptr(dbeta), _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _stream()), 'bn_bwd_finalize_raw')
  else:
    blocks = L().asm_bn_stats_blocks(M, Cn)
    part = empty((blocks, 2, Cn), F32, dy)
    check(L().asm_bn_bwd_reduce(_ptr(dy), _ptr(x), _ptr(yout if relu else None), rk, M, Cn,
                                _ptr(mean), _ptr(invstd), _ptr(part), _stream()), 'bn_bwd_reduce')
    part = _compact(part, Cn)
    blocks = part.shape[0]
    check(L().asm_bn_bwd_finalize(_ptr(part), blocks, M, Cn, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                                  _ptr(dbeta), _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _stream()), 'bn_bwd_finalize')
  dx = torch.empty_like(x)
  dz = torch.empty_like(x) if want_dz else None
  check(L().asm_bn_bwd_apply(_ptr(dy), _ptr(x), _ptr(yout if relu else None), rk, M, Cn,
                             _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _ptr(dx), _ptr(dz), _stream()), 'bn_bwd_apply')
  if ev is not None:
    ev.record()
  return dx, dz


def bn_apply_dual(xa, xb, M, Cn, scale_a, shift_a, scale_b, shift_b, relu, want_mask=False):
  """[relu](bn_a(xa) + bf16(bn_b(xb))) in one pass -> y, or (y, packed ReLU mask) with want_mask"""
  y = torch.empty_like(xa)
  mask = empty((M, Cn // 8), torch.uint8, xa) if (want_mask and relu) else None
  ev = _bn_ev(M * Cn * (6.0 + (0.125 if mask is not None else 0.0)))
  check(L().asm_bn_apply2(_ptr(xa), _ptr(xb), _ptr(y), M, Cn, _ptr(scale_a), _ptr(shift_a), _ptr(scale_b), _ptr(shift_b),
                          1 if relu else 0, _ptr(mask), _stream()), 'bn_apply2')
  if ev is not None:
    ev.record()
  return (y, mask) if want_mask else y


def bn_bwd_dual(dy, xa, xb, mask, M, Cn, bn_a, bn_b):
  """Backward of out = relu(bn_a(xa) + bn_b(xb)) from the un-masked gradient dy and the packed ReLU mask of out, both
  batch norms in one reduce + one apply.  bn_x = (gamma, mean, invstd, dgamma, dbeta) -> (dxa, dxb)."""
  ev = _bn_ev(M * Cn * (2 * 6.0 + 4.0 + 2 * 0.125))
  blocks = L().asm_bn_stats_blocks(M, Cn)
  pa, pb = empty((blocks, 2, Cn), F32, dy), empty((blocks, 2, Cn), F32, dy)
  check(L().asm_bn_bwd_reduce2(_ptr(dy), _ptr(xa), _ptr(xb), _ptr(mask), M, Cn, _ptr(bn_a[1]), _ptr(bn_a[2]),
                               _ptr(bn_b[1]), _ptr(bn_b[2]), _ptr(pa), _ptr(pb), _stream()), 'bn_bwd_reduce2')
  co = empty((6, Cn), F32, dy)
  for i, (part, (gamma, mean, invstd, dgamma, dbeta)) in enumerate(((pa, bn_a), (pb, bn_b))):
    part = _compact(part, Cn)
    check(L().asm_bn_bwd_finalize(_ptr(part), part.shape[0], M, Cn, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                                  _ptr(dbeta), _ptr(co[3 * i]), _ptr(co[3 * i + 1]), _ptr(co[3 * i + 2]), _stream()),
          'bn_bwd_finalize')
  dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
  check(L().asm_bn_bwd_apply2(_ptr(dy), _ptr(xa), _ptr(xb), _ptr(mask), M, Cn, _ptr(co), _ptr(dxa), _ptr(dxb), _stream()),
        'bn_bwd_apply2')
  if ev is not None:
    ev.record()
  return dxa, dxb


_SMALL_BN_ROWS = None


def bn_small_ok(M: int) -> bool:
  global _SMALL_BN_ROWS
  if _SMALL_BN_ROWS is None:
    _SMALL_BN_ROWS = int(L().asm_bn_small_max_rows())
  return M <= _SMALL_BN_ROWS


def bn_small_fwd(x, M, Cn, gamma, beta, eps, momentum, mm, mv, relu, want_mask):
  """Whole training-mode BN of a small tensor in one launch -> (y, mask or None, mean, invstd)."""
  y = torch.empty_like(x)
  co = empty((2, Cn), F32, x)
  mask = empty((M, Cn // 8), torch.uint8, x) if (want_mask and relu) else None
  check(L().asm_bn_small_fwd(_ptr(x), _ptr(y), M, Cn, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(mm), _ptr(mv),
                             _ptr(co[0]), _ptr(co[1]), 1 if relu else 0, _ptr(mask), _stream()), 'bn_small_fwd')
  return y, mask, co[0], co[1]


def bn_small_bwd(dy, x, mask, M, Cn, gamma, mean, invstd, dgamma, dbeta):
  dx = torch.empty_like(x)
  check(L().asm_bn_small_bwd(_ptr(dy), _ptr(x), _ptr(mask), M, Cn, _ptr(gamma), _ptr(mean), _ptr(invstd),
                             _ptr(dgamma), _ptr(dbeta), _ptr(dx), _stream()), 'bn_small_bwd')
  return dx


# ---------------------------------------------------------------------------------------------------
# pooling / resampling
# ---------------------------------------------------------------------------------------------------
def maxpool3x3s2_fwd(x):
  N, H, W, Cn = x.shape
  Ho, Wo = (H + 1) // 2, (W + 1) // 2
  y = empty((N, Ho, Wo, Cn), BF16, x)
  am = empty((N, Ho, Wo, Cn), torch.uint8, x)
  check(L().asm_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(am), N, H, W, Cn, _stream()), 'maxpool_fwd')
  return y, am


def maxpool3x3s2_bwd(dy, am, in_shape):
  N, H, W, Cn = in_shape
  dx = empty(in_shape, BF16, dy)
  check(L().asm_maxpool3x3s2_bwd(_ptr(dy), _ptr(am), _ptr(dx), N, H, W, Cn, _stream()), 'maxpool_bwd')
  return dx


def avgpool_fwd(x, k, stride, pad, Ho, Wo, count_valid):
  N, H, W, Cn = x.shape
  y = empty((N, Ho, Wo, Cn), BF16, x)
  check(L().asm_avgpool_fwd(_ptr(x), _ptr(y), N, H, W, Cn, k, stride, pad, Ho, Wo, 1 if count_valid else 0,
                            _stream()), 'avgpool_fwd')
  return y


def avgpool_bwd(dy, in_shape, k, stride, pad, count_valid, addend=None):
  """-> dx (+ addend when given; the sum is written into the addend's buffer when ``addend`` is passed)."""
  N, H, W, Cn = in_shape
  dx = addend if addend is not None else empty(in_shape, BF16, dy)
  check(L().asm_avgpool_bwd(_ptr(dy), _ptr(dx), N, H, W, Cn, k, stride, pad, dy.shape[1], dy.shape[2],
                            1 if count_valid else 0, _ptr(addend), _stream()), 'avgpool_bwd')
  return dx


def upsample2x_bwd(dy, mask=None):
  """2x2 block sums of dy [* mask bit] (mask: packed ReLU mask [N*H*W, C/8] of the full-resolution tensor)"""
  N, H, W, Cn = dy.shape
  dx = empty((N, H // 2, W // 2, Cn), BF16, dy)
  if mask is not None:
    check(L().asm_upsample2x_bwd_masked(_ptr(dy), _ptr(mask), _ptr(dx), N, H // 2, W // 2, Cn, _stream()),
          'upsample2x_bwd_masked')
  else:
    check(L().asm_upsample2x_bwd(_ptr(dy), _ptr(dx), N, H // 2, W // 2, Cn, _stream()), 'upsample2x_bwd')
  return dx


def blur_out_size(n, k, stride):
  return (n + 2 * ((k - 1) // 2) - k) // stride + 1


def blurpool_fwd(x, k, stride):
  N, H, W, Cn = x.shape
  y = empty((N, blur_out_size(H, k, stride), blur_out_size(W, k, stride), Cn), BF16, x)
  check(L().asm_blurpool_fwd(_ptr(x), _ptr(y), N, H, W, Cn, k, stride, _stream()), 'blurpool_fwd')
  return y


def blurpool_bwd(dy, in_shape, k, stride):
  N, H, W, Cn = in_shape
  dx = empty(in_shape, BF16, dy)
  check(L().asm_blurpool_bwd(_ptr(dy), _ptr(dx), N, H, W, Cn, k, stride, _stream()), 'blurpool_bwd')
  return dx


def gap_fwd(x):
  N, H, W, Cn = x.shape
  y = empty((N, 1, 1, Cn), BF16, x)
  check(L().asm_gap_fwd(_ptr(x), _ptr(y), N, H * W, Cn, _stream()), 'gap_fwd')
  return y


def gap_bwd(dy, in_shape):
  N, H, W, Cn = in_shape
  dx = empty(in_shape, BF16, dy)
  check(L().asm_gap_bwd(_ptr(dy), _ptr(dx), N, H * W, Cn, _stream()), 'gap_bwd')
  return dx


# ---------------------------------------------------------------------------------------------------
# SK / SE
# ---------------------------------------------------------------------------------------------------
def sk_gap(f, F_):
  N, H, W, _ = f.shape
  s = empty((N, 1, 1, F_), BF16, f)
  check(L().asm_sk_gap(_ptr(f), _ptr(s), N, H * W, F_, _stream()), 'sk_gap')
  return s


def sk_select_fwd(f, att, F_):
  N, H, W, _ = f.shape
  v = empty((N, H, W, F_), BF16, f)
  check(L().asm_sk_select_fwd(_ptr(f), _ptr(att), _ptr(v), N, H * W, F_, _stream()), 'sk_select_fwd')
  return v


def sk_select_bwd_att(f, dv, att, F_):
  N, H, W, _ = f.shape
  datt = empty((N, 1, 1, 2 * F_), BF16, f)
  check(L().asm_sk_select_bwd_att(_ptr(f), _ptr(dv), _ptr(att), _ptr(datt), N, H * W, F_, _stream()),
        'sk_select_bwd_att')
  return datt


def sk_select_bwd_f(dv, att, ds, F_):
  N, H, W, _ = dv.shape
  df = empty((N, H, W, 2 * F_), BF16, dv)
  check(L().asm_sk_select_bwd_f(_ptr(dv), _ptr(att), _ptr(ds), _ptr(df), N, H * W, F_, _stream()), 'sk_select_bwd_f')
  return df


# fused form: the 3x3 convolution's batch norm + ReLU applied on the fly (y = conv output [N,H,W,2F], never normalised in HBM)
def sk_gap_bn(y, scale, shift, F_, mean=None, invstd=None):
  """-> s, or with (mean, invstd): (s, per-image mask statistics [N, 2, 2F] f32 for the factorised BN backward)"""
  N, H, W, _ = y.shape
  s = empty((N, 1, 1, F_), BF16, y)
  if mean is not None:
    st = empty((N, 2, 2 * F_), F32, y)
    check(L().asm_sk_gap_bn_stats(_ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(s), _ptr(st), N, H * W,
                                  F_, _stream()), 'sk_gap_bn_stats')
    return s, st
  check(L().asm_sk_gap_bn(_ptr(y), _ptr(scale), _ptr(shift), _ptr(s), N, H * W, F_, _stream()), 'sk_gap_bn')
  return s


def sk_select_bn_fwd(y, scale, shift, att, F_):
  N, H, W, _ = y.shape
  v = empty((N, H, W, F_), BF16, y)
  check(L().asm_sk_select_bn_fwd(_ptr(y), _ptr(scale), _ptr(shift), _ptr(att), _ptr(v), N, H * W, F_, _stream()),
        'sk_select_bn_fwd')
  return v


def sk_select_bn_bwd_att(y, scale, shift, dv, att, F_, mean=None, invstd=None):
  """-> datt, or with (mean, invstd): (datt, per-image gradient statistics [N, 2, 2F] f32)"""
  N, H, W, _ = y.shape
  datt = empty((N, 1, 1, 2 * F_), BF16, y)
  if mean is not None:
    st = empty((N, 2, 2 * F_), F32, y)
    check(L().asm_sk_select_bn_bwd_att_stats(_ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), _ptr(dv), _ptr(att),
                                             _ptr(datt), _ptr(st), N, H * W, F_, _stream()), 'sk_select_bn_bwd_att_stats')
    return datt, st
  check(L().asm_sk_select_bn_bwd_att(_ptr(y), _ptr(scale), _ptr(shift), _ptr(dv), _ptr(att), _ptr(datt), N, H * W, F_,
                                     _stream()), 'sk_select_bn_bwd_att')
  return datt


def sk_bn_bwd(dv, att, ds, y, scale, shift, gamma, mean, invstd, dgamma, dbeta, F_, grad_stats=None, mask_stats=None):
  """BN backward of the SK unit's 2F-channel batch norm from dV (df is rebuilt in registers) -> dy [N,H,W,2F].
  With the per-image statistics of sk_select_bn_bwd_att / sk_gap_bn the reduce pass is replaced by a tiny finalize."""
  N, H, W, C2 = y.shape
  HW, M = H * W, N * H * W
  if grad_stats is not None and mask_stats is not None:
    ev = _bn_ev(M * (2.0 * C2 + 2.0 * F_ + 2.0 * C2))         # one pass over (y, dV), one write of dy
    co = empty((3, C2), F32, y)
    check(L().asm_sk_bn_bwd_finalize(_ptr(grad_stats), _ptr(mask_stats), _ptr(att), _ptr(ds), N, HW, F_, _ptr(gamma),
                                     _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(co[0]), _ptr(co[1]),
                                     _ptr(co[2]), _stream()), 'sk_bn_bwd_finalize')
    dy = torch.empty_like(y)
    check(L().asm_sk_bn_bwd_apply(_ptr(dv), _ptr(att), _ptr(ds), _ptr(y), _ptr(scale), _ptr(shift), _ptr(co[0]),
                                  _ptr(co[1]), _ptr(co[2]), _ptr(dy), N, HW, F_, _stream()), 'sk_bn_bwd_apply')
    if ev is not None:
      ev.record()
    return dy
  blocks = L().asm_sk_bn_bwd_blocks(N, HW, F_)
  if blocks <= 0:
    raise ValueError('sk_bn_bwd: bad shape')
  ev = _bn_ev(M * (2 * (2.0 * C2 + 2.0 * F_) + 2.0 * C2))     # two passes over (y, dV), one write of dy
  part = empty((blocks, 2, C2), F32, y)
  check(L().asm_sk_bn_bwd_reduce(_ptr(dv), _ptr(att), _ptr(ds), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean),
                                 _ptr(invstd), N, HW, F_, _ptr(part), _stream()), 'sk_bn_bwd_reduce')
  part = _compact(part, C2)
  co = empty((3, C2), F32, y)
  check(L().asm_bn_bwd_finalize(_ptr(part), part.shape[0], M, C2, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                                _ptr(dbeta), _ptr(co[0]), _ptr(co[1]), _ptr(co[2]), _stream()), 'bn_bwd_finalize')
  dy = torch.empty_like(y)
  check(L().asm_sk_bn_bwd_apply(_ptr(dv), _ptr(att), _ptr(ds), _ptr(y), _ptr(scale), _ptr(shift), _ptr(co[0]),
                                _ptr(co[1]), _ptr(co[2]), _ptr(dy), N, HW, F_, _stream()), 'sk_bn_bwd_apply')
  if ev is not None:
    ev.record()
  return dy


def se_scale_fwd(x, e):
  N, H, W, Cn = x.shape
  y = torch.empty_like(x)
  check(L().asm_se_scale_fwd(_ptr(x), _ptr(e), _ptr(y), N, H * W, Cn, _stream()), 'se_scale_fwd')
  return y


def se_scale_bwd_e(x, dy, e):
  N, H, W, Cn = x.shape
  de = empty((N, 1, 1, Cn), BF16, x)
  check(L().asm_se_scale_bwd_e(_ptr(x), _ptr(dy), _ptr(e), _ptr(de), N, H * W, Cn, _stream()), 'se_scale_bwd_e')
  return de


def se_scale_bwd_x(dy, e, dsq):
  N, H, W, Cn = dy.shape
  dx = torch.empty_like(dy)
  check(L().asm_se_scale_bwd_x(_ptr(dy), _ptr(e), _ptr(dsq), _ptr(dx), N, H * W, Cn, _stream()), 'se_scale_bwd_x')
  return dx


# ---------------------------------------------------------------------------------------------------
# element-wise / loss / input / optimiser
# ---------------------------------------------------------------------------------------------------
def relu_fwd(x):
  y = torch.empty_like(x)
  check(L().asm_relu_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), 'relu_fwd')
  return y


def relu_bwd(dy, y):
  dx = torch.empty_like(dy)
  check(L().asm_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()), 'relu_bwd')
  return dx


def mask_apply(dy, mask):
  dx = torch.empty_like(dy)
  check(L().asm_mask_apply(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), _stream()), 'mask_apply')
  return dx


def add_bf16(a, b, out=None):
  if out is None:
    out = torch.empty_like(a)
  check(L().asm_add_bf16(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), 'add_bf16')
  return out


def bias_add_f32(y, bias, M, Cn, ldy):
  check(L().asm_bias_add_f32(_ptr(y), _ptr(bias), M, Cn, ldy, _stream()), 'bias_add')


def bias_grad_bf16(dz, M, Cn, ld, dbias):
  check(L().asm_bias_grad_bf16(_ptr(dz), M, Cn, ld, _ptr(dbias), _stream()), 'bias_grad')


def cast_f32_to_bf16(x, y):
  check(L().asm_cast_f32_to_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), 'cast')


def cast_bf16_to_f32(x, y):
  check(L().asm_cast_bf16_to_f32(_ptr(x), _ptr(y), x.numel(), _stream()), 'cast_bf16_to_f32')


def softmax_ce(logits, ld, targets, teacher, B, Cn, label_smoothing, kd_temp, loss_scale, ld_out, want_grad=True):
  loss_rows = empty((B,), F32, logits)
  dlogits = empty((B, 1, 1, ld_out), BF16, logits) if want_grad else None
  check(L().asm_softmax_ce(_ptr(logits), ld, _ptr(targets), _ptr(teacher), B, Cn, label_smoothing, kd_temp,
                           loss_scale, _ptr(loss_rows), _ptr(dlogits), ld_out, _stream()), 'softmax_ce')
  return loss_rows, dlogits


def onehot(labels_i32, B, Cn):
  out = empty((B, Cn), F32, labels_i32)
  check(L().asm_onehot(_ptr(labels_i32), _ptr(out), B, Cn, _stream()), 'onehot')
  return out


def softmax_rows(x, B, Cn, inv_temp):
  y = torch.empty_like(x)
  check(L().asm_softmax_rows(_ptr(x), _ptr(y), B, Cn, inv_temp, _stream()), 'softmax_rows')
  return y


def mean_f32(x):
  out = empty((1,), F32, x)
  check(L().asm_mean_f32(_ptr(x), x.numel(), _ptr(out), _stream()), 'mean')
  return out


def mixup_meansub(images, mixup_type, lam1, lam2):
  Bin, H, W, c = images.shape
  if c != 3:
    raise ValueError('images must be [B,H,W,3]')
  if images.dtype == torch.uint8:
    is_u8 = 1
  elif images.dtype == F32:
    is_u8 = 0
  else:
    raise ValueError('images must be uint8 or float32 (0..255)')
  Bout = Bin // 2 if mixup_type == 1 else Bin
  out = empty((Bout, H + 6, W + 6, 4), BF16, images)
  check(L().asm_mixup_meansub(_ptr(images), is_u8, Bin, H, W, mixup_type, _ptr(lam1), _ptr(lam2), _ptr(out),
                              _stream()), 'mixup_meansub')
  return out


def mixup_labels(y, mixup_type, lam1, lam2):
  Bin, Cn = y.shape
  Bout = Bin // 2 if mixup_type == 1 else Bin
  out = empty((Bout, Cn), F32, y)
  check(L().asm_mixup_labels(_ptr(y), Bin, Cn, mixup_type, _ptr(lam1), _ptr(lam2), _ptr(out), _stream()),
        'mixup_labels')
  return out


def sgd_momentum(w, accum, grad, w_bf16, lr, momentum, weight_decay, grad_scale):
  check(L().asm_sgd_momentum(_ptr(w), _ptr(accum), _ptr(grad), _ptr(w_bf16), w.numel(), lr, momentum, weight_decay,
                             grad_scale, _stream()), 'sgd_momentum')


# ---------------------------------------------------------------------------------------------------
# sigmoid loss, GeM, DropBlock, evaluation metrics
# ---------------------------------------------------------------------------------------------------
def sigmoid_ce(logits, ld, targets, B, Cn, loss_scale, ld_out, want_grad=True):
  """-> (loss_out [2] = {loss, sum(onehot)}, dlogits bf16 [B,1,1,ld_out])"""
  rows = empty((B, 2), F32, logits)
  out = empty((2,), F32, logits)
  dz = empty((B, 1, 1, ld_out), BF16, logits) if want_grad else None
  check(L().asm_sigmoid_ce(_ptr(logits), ld, _ptr(targets), B, Cn, loss_scale, _ptr(rows), _ptr(out), _ptr(dz), ld_out,
                           _stream()), 'sigmoid_ce')
  return out, dz


def gem_fwd(x, p=3.0):
  N, H, W, Cn = x.shape
  y = empty((N, 1, 1, Cn), BF16, x)
  ssum = empty((N, Cn), F32, x)
  check(L().asm_gem_fwd(_ptr(x), _ptr(y), _ptr(ssum), N, H * W, Cn, p, _stream()), 'gem_fwd')
  return y, ssum


def gem_bwd(x, dy, ssum, p=3.0):
  N, H, W, Cn = x.shape
  dx = torch.empty_like(x)
  check(L().asm_gem_bwd(_ptr(x), _ptr(dy), _ptr(ssum), _ptr(dx), N, H * W, Cn, p, _stream()), 'gem_bwd')
  return dx


def dropblock_mask(uniform, gamma, H, W, Cn, block_size, gamma_dev=None):
  """gamma_dev: float32 device tensor [1] holding the Bernoulli mean instead of the host scalar (a recorded step replays
  the launch with its recorded arguments; the host rewrites gamma_dev[0] as keep_prob follows its schedule)."""
  keep = empty((H, W, Cn), F32, uniform)
  scale = empty((1,), F32, uniform)
  if gamma_dev is not None:
    check(L().asm_dropblock_mask_dev(_ptr(uniform), _ptr(gamma_dev), H, W, Cn, block_size, _ptr(keep), _ptr(scale),
                                     _stream()), 'dropblock_mask_dev')
  else:
    check(L().asm_dropblock_mask(_ptr(uniform), gamma, H, W, Cn, block_size, _ptr(keep), _ptr(scale), _stream()),
          'dropblock_mask')
  return keep, scale


def memcpy(dst: torch.Tensor, src: torch.Tensor):
  """dst <- src, two contiguous device tensors of the same byte size, through the library (asm_memcpy_async: seen by a
  launch tape that is being recorded, unlike a framework copy kernel)."""
  nb = src.numel() * src.element_size()
  if not (dst.is_contiguous() and src.is_contiguous()) or dst.numel() * dst.element_size() != nb:
    raise ValueError('memcpy needs two contiguous tensors of the same byte size')
  check(L().asm_memcpy_async(_ptr(dst), _ptr(src), nb, _stream()), 'memcpy_async')
  return dst


def dropblock_apply(x, keep, scale, relu=False, relu_mask_from=None):
  N = x.shape[0]
  y = torch.empty_like(x)
  check(L().asm_dropblock_apply(_ptr(x), _ptr(keep), _ptr(scale), _ptr(relu_mask_from), 1 if relu else 0, _ptr(y), N,
                                x.numel() // N, _stream()), 'dropblock_apply')
  return y


def eval_rows(logits, ld, labels_i32, B, Cn):
  pred = empty((B,), torch.int32, logits)
  vals = empty((3, B), F32, logits)
  check(L().asm_eval_rows(_ptr(logits), ld, _ptr(labels_i32), B, Cn, _ptr(pred), _ptr(vals[0]), _ptr(vals[1]),
                          _ptr(vals[2]), _stream()), 'eval_rows')
  return pred, vals[0], vals[1], vals[2]


def eval_accumulate(conf, top1, top5, state33):
  check(L().asm_eval_accumulate(_ptr(conf), _ptr(top1), _ptr(top5), conf.numel(), _ptr(state33), _stream()),
        'eval_accumulate')


def resize_crop_flip(src_u8, descs_dev, n, out_h, out_w, subtract_mean):
  """src_u8: 1-D uint8 device buffer with the decoded images back to back; descs_dev: uint8 device view of
  n packed struct asm_image_desc (56 bytes each).  Returns float32 [n, out_h, out_w, 3]."""
  if descs_dev.numel() != 56 * n:
    raise ValueError('descriptor table must hold %d bytes' % (56 * n))
  out = torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=src_u8.device)
  if n:
    check(L().asm_resize_crop_flip(_ptr(src_u8), src_u8.numel(), _ptr(descs_dev), n, out_h, out_w,
                                   1 if subtract_mean else 0, _ptr(out), _stream()), 'resize_crop_flip')
  return out
