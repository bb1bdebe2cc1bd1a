#!/usr/bin/env python
"""GPU debug helper: dump every teacher-forced backward comparison (tests/model_parity.py) sorted by error/tolerance."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assembled_cnn_amd import lib, ops  # noqa: E402
from tests import model_parity as mp  # noqa: E402

name, batch, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kp = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
ops.set_library(None, is_double=False)
lib.load()
errs, st = mp.check_teacher_forced_backward(name, 'cuda', batch, size, keep_prob=kp, dx_tol=1, lazy_tol=1, dparam_tol=1,
                                            dw_tol=1, squeeze_tol=10, sk_tol=1)
print(st)
for e in sorted(errs, key=lambda t: -t[2]):
  print('%-78s %-14s %.3e %s' % (e[0][-78:], e[1], e[2], e[3]))
