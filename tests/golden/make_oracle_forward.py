#!/usr/bin/env python
"""Compute and commit the ORACLE side of the literal-size forward tests (tests/test_gpu_model.py: BASELINE configs 1, 3, 4, 5).

The oracle (oracle/assembled_oracle.py, pinned to the reference by tests/golden/make_reference_taps.py / _step.py) is run on
the tests' own seeded inputs; what is stored per configuration is tests/golden/oracle_forward/<key>.npz: logits, loss, the
bf16-vs-fp32 noise figure where the test calibrates on it, and a 65536-element subset of every named tap.  The tests load
these instead of spending 1 - 2 minutes of CPU oracle each on the GPU box (ASM_ORACLE_LIVE=1 recomputes there).
usage: python tests/golden/make_oracle_forward.py [key ...]      (CPU only; tens of minutes for all four)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from assembled_cnn_amd import ops  # noqa: E402
from tests import model_parity as mp  # noqa: E402
from tests.cpu_double import CpuDouble  # noqa: E402


def main():
  ops.set_library(CpuDouble(), is_double=True)    # the product side of the harness is not run, but building the pair needs a library
  os.environ['ASM_ORACLE_LIVE'] = '1'
  os.makedirs(mp.GOLDEN_FWD, exist_ok=True)
  want = set(sys.argv[1:])
  jobs = {
      'config1_r50v1_eval_b64_224': lambda: fwd('r50v1', 64, 224, False),
      'config3_a-r50-d_train_b256_224': lambda: fwd('a-r50-d', 256, 224, True),
      'config4_a-r50-d_mixup1_ls_512in_224': lambda: train_fwd('a-r50-d', 512, 224, mixup_type=1, label_smoothing=0.1),
      'config5_a-r152_kd_b128_224': lambda: train_fwd('a-r152', 128, 224, kd_temp=1.0, noise_floor=True),
      'noise_floor_a-r152_b8_128': lambda: mp.oracle_noise_floor_record(mp.make_pair('a-r152', 'cpu', 2, 64)[0], 'a-r152', 8, 128),
  }
  for key, job in jobs.items():
    if want and key not in want:
      continue
    rec = job()
    np.savez_compressed(os.path.join(mp.GOLDEN_FWD, key + '.npz'), **rec)
    print(key, {k: (v.shape if hasattr(v, 'shape') and v.shape else float(v)) for k, v in rec.items()}, flush=True)


def fwd(name, batch, size, training):
  from tests import util
  om, _ = mp.make_pair(name, 'cpu', 2, 64)     # the oracle's variables do not depend on the probe size
  if not training:
    util.perturb_bn_state(om, 7)
  return mp.oracle_forward_record(om, name, batch, size, training)


def train_fwd(name, n_in, size, **kw):
  return mp.oracle_train_forward_record(name, n_in, size, **kw)


if __name__ == '__main__':
  torch.manual_seed(0)
  main()
