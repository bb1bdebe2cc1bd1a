"""ctypes binding of libasm_hip.so (the C ABI declared in include/asm_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ASM_HIP_LIB=<path>: load another build of the same ABI (same-box A/B runs of a kernel change; tools/ab/)
LIB_PATH = os.environ.get('ASM_HIP_LIB') or os.path.join(HERE, 'libasm_hip.so')

ASM_OK, ASM_EINVAL, ASM_ENOTSUP, ASM_EHIP = 0, -1, -2, -3
ABI_VERSION = 5


class AsmError(RuntimeError):
  pass


class ImageDesc(C.Structure):
  """struct asm_image_desc"""
  _fields_ = [('src_offset', C.c_int64), ('Hs', C.c_int32), ('Ws', C.c_int32),
              ('crop_y', C.c_int32), ('crop_x', C.c_int32), ('crop_h', C.c_int32), ('crop_w', C.c_int32),
              ('resize_h', C.c_int32), ('resize_w', C.c_int32), ('out_y', C.c_int32), ('out_x', C.c_int32),
              ('flip', C.c_int32), ('reserved', C.c_int32)]


class ConvDesc(C.Structure):
  """struct asm_conv_desc"""
  _fields_ = [('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32),
              ('K', C.c_int32), ('R', C.c_int32), ('S', C.c_int32),
              ('stride', C.c_int32), ('pad', C.c_int32),
              ('Ho', C.c_int32), ('Wo', C.c_int32),
              ('x_img_pitch', C.c_int64), ('x_row_pitch', C.c_int32), ('x_pix_pitch', C.c_int32),
              ('ldy', C.c_int32), ('out_f32', C.c_int32)]


class ModelCfg(C.Structure):
  """struct asm_model_cfg"""
  _fields_ = [(n, C.c_int32) for n in (
      'resnet_size', 'resnet_version', 'num_classes', 'use_se_block', 'use_sk_block', 'use_resnet_d',
      'anti_alias_filter_size', 'anti_alias_type', 'bl_alpha', 'bl_beta', 'zero_gamma', 'no_downsample', 'pool_type',
      'embedding_size', 'dtype', 'mixup_type')] + [(n, C.c_float) for n in (
          'bn_momentum', 'bn_eps', 'loss_scale', 'label_smoothing', 'kd_temp', 'weight_decay', 'momentum')]


class PlanEntry(C.Structure):
  """struct asm_plan_entry"""
  _fields_ = [(n, C.c_int32) for n in ('kind', 'N', 'H', 'W', 'C', 'K', 'R', 'S', 'stride', 'Ho', 'Wo', 'flags',
                                       'trainable', 'reserved')] + [
      ('param_offset', C.c_int64), ('param_elems', C.c_int64), ('name', C.c_char * 112)]


class PlanSummary(C.Structure):
  """struct asm_plan_summary"""
  _fields_ = [('n_entries', C.c_int32), ('trainable_tensors', C.c_int32), ('trainable_elems', C.c_int64),
              ('forward_macs_per_image', C.c_int64), ('wgrad_workspace_bytes', C.c_int64)]


class Tuning(C.Structure):
  """struct asm_tuning: the kernel-selection overrides (the library itself reads no environment variable)"""
  _fields_ = [(n, C.c_int32) for n in (
      'igemm_mode', 'igemm_tile', 'igemm_pfa', 'dgrad_parity', 'wgrad_halo', 'wgrad_big', 'wgrad_splits', 'bn_rows', 'igemm3',
      'gemm1', 'wgrad_ring', 'igemm8')]


# environment variable of the HOST -> asm_tuning field (same-box A/B runs, tests); unset = the library's default
TUNING_ENV = {'ASM_IGEMM_MODE': 'igemm_mode', 'ASM_IGEMM_TILE': 'igemm_tile', 'ASM_IGEMM_PFA': 'igemm_pfa',
              'ASM_DGRAD_PARITY': 'dgrad_parity', 'ASM_WGRAD_HALO': 'wgrad_halo', 'ASM_WGRAD_BIG': 'wgrad_big',
              'ASM_WGRAD_SPLITS': 'wgrad_splits', 'ASM_BN_ROWS': 'bn_rows', 'ASM_IGEMM3': 'igemm3', 'ASM_GEMM1': 'gemm1',
              'ASM_WGRAD_RING': 'wgrad_ring', 'ASM_IGEMM8': 'igemm8'}


def apply_env_tuning(lib) -> 'Tuning':
  """asm_set_tuning(defaults overridden by whichever ASM_* variables are set in this process's environment)"""
  t = Tuning()
  lib.asm_tuning_defaults(C.byref(t))
  for env, field in TUNING_ENV.items():
    v = os.environ.get(env)
    if v not in (None, ''):
      setattr(t, field, int(v))
  check_code = lib.asm_set_tuning(C.byref(t))
  if check_code != ASM_OK:
    raise ValueError('asm_set_tuning rejected the ASM_* overrides: %s' % (lib.asm_last_error() or b'').decode())
  return t


ASM_F32, ASM_BF16, ASM_F16 = 0, 1, 2
ASM_AA_SCONV, ASM_AA_PROJ = 1, 2
POOL_TYPES = {'gap': 0, 'gem': 1, 'flatten': 2}
(PLAN_CONV, PLAN_BN, PLAN_DENSE, PLAN_MAXPOOL, PLAN_AVGPOOL, PLAN_BLURPOOL, PLAN_GAP, PLAN_GEM, PLAN_FLATTEN, PLAN_SK_GAP,
 PLAN_SK_SELECT, PLAN_SE_SCALE, PLAN_ADD) = range(13)

_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol declared in include/asm_hip.h
SIGNATURES = {
    'asm_last_error': (C.c_char_p, []),
    'asm_abi_version': (_I, []),
    'asm_launch_count': (C.c_ulonglong, []),
    'asm_stream_join': (_I, [_P, _P]),
    'asm_memcpy_async': (_I, [_P, _P, _Z, _P]),
    'asm_allreduce_bucket': (_I, [_P, _Z, _I, _P, _P, _P]),
    'asm_tape_begin': (_I, []),
    'asm_tape_mark': (_I, []),
    'asm_tape_end': (_I, []),
    'asm_tape_info': (_I, [_I, C.POINTER(C.c_int64 * 6)]),
    'asm_tape_replay': (_I, [_I, _I]),
    'asm_tape_free': (_I, [_I]),
    'asm_tuning_defaults': (None, [C.POINTER(Tuning)]),
    'asm_set_tuning': (_I, [C.POINTER(Tuning)]),
    'asm_get_tuning': (None, [C.POINTER(Tuning)]),
    'asm_conv2d_fprop': (_I, [_D, _P, _P, _P, _P, _P]),
    'asm_conv2d_stats_blocks': (_I, [_D]),
    'asm_conv2d_dgrad': (_I, [_D, _P, _P, _P, _P, _P]),
    'asm_conv2d_dgrad_masked': (_I, [_D, _P, _P, _P, _P, _P, _P]),
    'asm_conv2d_dgrad_pooled': (_I, [_D, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'asm_conv2d_dgrad_bnred_blocks': (_I, [_D]),
    'asm_conv2d_dgrad_bnred': (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asm_conv2d_wgrad_workspace_bytes': (_Z, [_D]),
    'asm_conv2d_wgrad': (_I, [_D, _P, _P, _P, _P, _Z, _P]),
    'asm_filter_transpose': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'asm_filter_transpose_batched': (_I, [_P, _P, _P, _I, C.c_longlong, _P]),
    'asm_filter_transpose_tiled': (_I, [_P, _P, _P, _I, _I, _P]),
    'asm_stem_pack_filter': (_I, [_P, _P, _I, _I, _P]),
    'asm_stem_unpack_grad': (_I, [_P, _P, _I, _I, _P]),
    'asm_stem_pad_input': (_I, [_P, _I, _P, _I, _I, _I, _P]),
    'asm_bn_stats_blocks': (_I, [_I, _I]),
    'asm_bn_stats': (_I, [_P, _I, _I, _P, _P]),
    'asm_bn_partials_compact': (_I, [_P, _I, _I, _P, _I, _P]),
    'asm_bn_finalize': (_I, [_P, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'asm_bn_infer_coeffs': (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    'asm_bn_apply': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'asm_bn_bwd_reduce': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'asm_bn_bwd_finalize': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asm_bn_bwd_finalize_raw': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asm_bn_bwd_apply': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    'asm_maxpool3x3s2_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'asm_maxpool3x3s2_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'asm_avgpool_fwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'asm_avgpool_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'asm_upsample2x_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'asm_upsample2x_bwd_masked': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'asm_blurpool_fwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'asm_blurpool_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'asm_gap_fwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'asm_gap_bwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'asm_sk_gap': (_I, [_P, _P, _I, _I, _I, _P]),
    'asm_sk_select_fwd': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_select_bwd_att': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_select_bwd_f': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_gap_bn': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_select_bn_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_select_bn_bwd_att': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_bn_bwd_blocks': (_I, [_I, _I, _I]),
    'asm_sk_bn_bwd_reduce': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    'asm_sk_bn_bwd_apply': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_gap_bn_stats': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_select_bn_bwd_att_stats': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_sk_bn_bwd_finalize': (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asm_se_scale_fwd': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'asm_se_scale_bwd_e': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_se_scale_bwd_x': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'asm_relu_fwd': (_I, [_P, _P, _Z, _P]),
    'asm_relu_bwd': (_I, [_P, _P, _P, _Z, _P]),
    'asm_add_bf16': (_I, [_P, _P, _P, _Z, _P]),
    'asm_mask_apply': (_I, [_P, _P, _P, _Z, _P]),
    'asm_bias_add_f32': (_I, [_P, _P, _I, _I, _I, _P]),
    'asm_bias_grad_bf16': (_I, [_P, _I, _I, _I, _P, _P]),
    'asm_cast_f32_to_bf16': (_I, [_P, _P, _Z, _P]),
    'asm_cast_bf16_to_f32': (_I, [_P, _P, _Z, _P]),
    'asm_softmax_ce': (_I, [_P, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _I, _P]),
    'asm_onehot': (_I, [_P, _P, _I, _I, _P]),
    'asm_softmax_rows': (_I, [_P, _P, _I, _I, _F, _P]),
    'asm_mean_f32': (_I, [_P, _I, _P, _P]),
    'asm_mixup_meansub': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'asm_mixup_labels': (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    'asm_sigmoid_ce': (_I, [_P, _I, _P, _I, _I, _F, _P, _P, _P, _I, _P]),
    'asm_gem_fwd': (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    'asm_gem_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'asm_dropblock_mask': (_I, [_P, _F, _I, _I, _I, _I, _P, _P, _P]),
    'asm_dropblock_mask_dev': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'asm_dropblock_apply': (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _P]),
    'asm_eval_rows': (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    'asm_eval_accumulate': (_I, [_P, _P, _P, _I, _P, _P]),
    'asm_sgd_momentum': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _P]),
    'asm_conv2d_fprop_bn': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'asm_bn_apply2': (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    'asm_bn_bwd_reduce2': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'asm_bn_bwd_apply2': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    'asm_bn_small_max_rows': (_I, []),
    'asm_bn_small_fwd': (_I, [_P, _P, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P, _P]),
    'asm_bn_small_bwd': (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'asm_dense_small': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P]),
    'asm_dense_small_wgrad': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P]),
    'asm_resize_crop_flip': (_I, [_P, C.c_int64, _P, _I, _I, _I, _I, _P, _P]),
    'asm_model_plan': (_I, [C.POINTER(ModelCfg), _I, _I, _I, C.POINTER(PlanEntry), _I, C.POINTER(PlanSummary)]),
}

# test-only entry points (include/asm_hip_debug.h): bound so tests can call them with full-width pointers, never
# called by the product package
DEBUG_SIGNATURES = {
    'asm_conv2d_fprop_naive': (_I, [_D, _P, _P, _P, _P]),
    'asm_conv2d_dgrad_naive': (_I, [_D, _P, _P, _P, _P]),
    'asm_conv2d_wgrad_naive': (_I, [_D, _P, _P, _P, _P]),
    'asm_debug_tr_probe': (_I, [_P, _P]),
    'asm_conv2d_wgrad_plan': (_I, [_D, C.POINTER(C.c_int32 * 6)]),
    'asm_debug_last_conv_kernel': (_I, []),
}

_lib = None


def load(path: str = LIB_PATH) -> C.CDLL:
  """Load the library and bind every declared symbol; raises if anything is missing."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(path):
    raise AsmError('%s not found -- build it with `python -m assembled_cnn_amd.build` '
                   '(there is no CPU fallback)' % path)
  # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; whichever HIP runtime is mapped first serves the whole
  # process.  Loaded before torch, this library would bind to the system ROCm runtime instead and the two would not share
  # devices, streams or allocations ("no ROCm-capable device is detected" at the first launch on a healthy GPU).
  import torch  # noqa: F401
  lib = C.CDLL(path)
  # the version check comes first: a stale build must report the ABI mismatch, not a missing symbol
  ver = getattr(lib, 'asm_abi_version', None)
  if ver is None:
    raise AsmError('%s does not export asm_abi_version: not a libasm_hip.so' % path)
  ver.restype, ver.argtypes = C.c_int, []
  if ver() != ABI_VERSION:
    raise AsmError('libasm_hip.so ABI version %d != %d (rebuild: python -m assembled_cnn_amd.build)' % (ver(), ABI_VERSION))
  for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
    fn = getattr(lib, name, None)
    if fn is None:
      raise AsmError('libasm_hip.so does not export %s' % name)
    fn.restype = res
    fn.argtypes = args
  apply_env_tuning(lib)
  _lib = lib
  return lib


CALLS = [0]   # C-ABI calls checked so far (bench.py reports launches per step from it)


def check(code: int, what: str = ''):
  CALLS[0] += 1
  if code != ASM_OK:
    msg = load().asm_last_error()
    msg = msg.decode() if msg else ''
    if code == ASM_EINVAL:
      raise ValueError('%s: %s' % (what, msg))
    if code == ASM_ENOTSUP:
      raise NotImplementedError('%s: %s' % (what, msg))
    raise AsmError('%s: HIP error: %s' % (what, msg))
