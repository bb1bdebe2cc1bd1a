"""The C-ABI shared library loads and exports every symbol include/asm_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header='asm_hip.h'):
  hdr = open(os.path.join(ROOT, 'include', header)).read()
  return sorted(set(re.findall(r'\b(asm_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
  import __graft_entry__
  __graft_entry__.build()
  from assembled_cnn_amd import lib
  so = ctypes.CDLL(lib.LIB_PATH)
  names = _declared()
  assert len(names) >= 50
  for n in names:
    assert hasattr(so, n), 'libasm_hip.so does not export %s' % n
  assert sorted(lib.SIGNATURES) == names, 'lib.py binding table and the header disagree'
  # test-only entry points live in their own header and binding table; the public header declares none of them
  dbg = _declared('asm_hip_debug.h')
  assert sorted(lib.DEBUG_SIGNATURES) == dbg and not set(dbg) & set(names)
  assert not [n for n in names if 'naive' in n or 'debug' in n]
  for n in dbg:
    assert hasattr(so, n), 'libasm_hip.so does not export %s' % n
  L = lib.load()
  assert L.asm_abi_version() == lib.ABI_VERSION


def test_struct_layout_matches_header():
  from assembled_cnn_amd.lib import ConvDesc
  # 11 int32 (44 B) + pad to 8 -> int64 at 48, then 4 int32 = 72 bytes
  assert ConvDesc.x_img_pitch.offset == 48
  assert ctypes.sizeof(ConvDesc) == 72


def test_product_has_no_cpu_path():
  """ops must refuse CPU tensors when the real library is bound (no silent fallback)."""
  import torch
  from assembled_cnn_amd import lib, ops
  ops.set_library(None, is_double=False)
  with pytest.raises(lib.AsmError):
    ops.relu_fwd(torch.zeros(8, dtype=torch.bfloat16))


def test_missing_library_fails_loudly(tmp_path):
  from assembled_cnn_amd import lib
  saved = lib._lib
  lib._lib = None
  try:
    with pytest.raises(lib.AsmError):
      lib.load(str(tmp_path / 'nope.so'))
  finally:
    lib._lib = saved


def test_product_never_imports_oracle():
  pkg = os.path.join(ROOT, 'assembled_cnn_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        src = open(os.path.join(dirpath, f)).read()
        assert 'import oracle' not in src and 'from oracle' not in src and 'cpu_double' not in src.replace(
            'double lives under tests/', ''), f


def test_launch_tape_bookkeeping_needs_no_gpu():
  """asm_tape_begin / mark / end / info / replay / free on an EMPTY recording: pure host code (csrc/tape.hip), so the
  segment bookkeeping and the error returns are checked here; recordings with launches are GPU tests."""
  from assembled_cnn_amd import lib
  L = lib.load()
  info = (ctypes.c_int64 * 6)()
  assert L.asm_tape_mark() < 0                       # nothing is being recorded
  assert L.asm_tape_end() < 0
  t = L.asm_tape_begin()
  assert t > 0
  assert L.asm_tape_begin() < 0, 'one recording per thread'
  assert b'already recording' in L.asm_last_error()
  assert L.asm_tape_replay(t, -1) < 0, 'a tape that is being recorded cannot be replayed'
  assert L.asm_tape_mark() == 1 and L.asm_tape_mark() == 2
  assert L.asm_tape_end() == t
  assert L.asm_tape_info(t, ctypes.byref(info)) == 0
  assert list(info) == [0, 0, 0, 0, 3, 0]            # no nodes; three (empty) segments
  for seg in (-1, 0, 1, 2):
    assert L.asm_tape_replay(t, seg) == 0
  assert L.asm_tape_replay(t, 3) < 0
  t2 = L.asm_tape_begin()
  assert t2 == t + 1 and L.asm_tape_end() == t2
  assert L.asm_tape_free(t) == 0 and L.asm_tape_free(t) < 0
  assert L.asm_tape_replay(t, -1) < 0 and L.asm_tape_info(t, ctypes.byref(info)) < 0
  assert L.asm_tape_free(t2) == 0
  # a recording abandoned by freeing it stops recording
  t3 = L.asm_tape_begin()
  assert L.asm_tape_free(t3) == 0
  assert L.asm_tape_end() < 0
  t4 = L.asm_tape_begin()
  assert t4 > 0 and L.asm_tape_end() == t4 and L.asm_tape_free(t4) == 0


def test_allreduce_bucket_argument_checks_need_no_gpu():
  """asm_allreduce_bucket (the gradient exchange for a C caller with its own ncclComm_t): the argument checks come before
  librccl is even resolved."""
  from assembled_cnn_amd import lib
  L = lib.load()
  fake = ctypes.c_void_p(0x1000)
  assert L.asm_allreduce_bucket(None, 16, lib.ASM_F32, fake, fake, fake) == lib.ASM_EINVAL
  assert L.asm_allreduce_bucket(fake, 16, lib.ASM_F32, None, fake, fake) == lib.ASM_EINVAL
  assert L.asm_allreduce_bucket(fake, 16, lib.ASM_F32, fake, None, fake) == lib.ASM_EINVAL and b'null stream' in L.asm_last_error()
  assert L.asm_allreduce_bucket(fake, 16, lib.ASM_F16, fake, fake, fake) == lib.ASM_ENOTSUP
  assert L.asm_allreduce_bucket(fake, 0, lib.ASM_BF16, fake, fake, fake) == lib.ASM_OK      # an empty bucket is nothing to do
