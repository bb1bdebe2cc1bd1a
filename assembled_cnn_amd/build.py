"""Build libasm_hip.so (gfx950 only) in-tree with hipcc.  `python -m assembled_cnn_amd.build`."""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB_PATH = os.path.join(HERE, 'libasm_hip.so')
SOURCES = ['conv_igemm.hip', 'conv_igemm8.hip', 'conv_gemm1.hip', 'conv_dgrad_s2.hip', 'conv_wgrad.hip', 'bn.hip', 'pool.hip', 'sk_se.hip', 'sk_fused.hip', 'dense_small.hip', 'misc.hip', 'extra.hip', 'input.hip', 'plan.hip', 'tape.hip', 'collective.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'igemm_common.h'), os.path.join(HERE, '..', 'include', 'asm_hip.h'),
           os.path.join(HERE, '..', 'include', 'asm_hip_debug.h')]
ARCH = 'gfx950'
CFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(exe):
    raise RuntimeError('hipcc not found: cannot build libasm_hip.so')
  return exe


def _stale(target: str, deps) -> bool:
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str) -> str:
  obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o')
  path = os.path.join(CSRC, src)
  if _stale(obj, [path] + HEADERS):
    cmd = [_hipcc()] + CFLAGS + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
  return obj


def build(force: bool = False, verbose: bool = False) -> str:
  os.makedirs(OBJ_DIR, exist_ok=True)
  if force:
    for f in os.listdir(OBJ_DIR):
      os.remove(os.path.join(OBJ_DIR, f))
  with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
    objs = list(ex.map(_compile, SOURCES))
  if force or _stale(LIB_PATH, objs):
    # link beside the target and rename over it: a process that has the old library mapped (this one, after lib.load())
    # keeps its inode; rewriting the file in place would pull the new bytes -- and the new embedded GPU code object --
    # under the loaded image (seen as "no ROCm-capable device is detected" at the next launch)
    tmp = LIB_PATH + '.tmp.%d' % os.getpid()
    cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      if os.path.exists(tmp):
        os.remove(tmp)
      raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    os.replace(tmp, LIB_PATH)
  if verbose:
    print('built', LIB_PATH)
  return LIB_PATH


if __name__ == '__main__':
  build(force='--force' in sys.argv, verbose=True)
