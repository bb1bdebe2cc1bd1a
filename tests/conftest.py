import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # The CPU oracle is most of the GPU tier's wall time, and on the GPU box's 256 hardware threads PyTorch's default (one
  # intra-op thread per hardware thread) is the slow choice for its layer-sized operators: one oracle-heavy test
  # (test_sk_blocks_forward_and_backward_at_batch_256[stage_4]) takes 33.2 s with the default, 20.5 s with 64 threads, 15.5 s
  # with 32, 16.7 s with 16 (round 6, same box).  OMP_NUM_THREADS in the environment wins.
  if 'OMP_NUM_THREADS' not in os.environ:
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.fixture
def cpu_double():
  """Install the CPU test double of the C ABI for the duration of a test (host-logic tests only)."""
  from assembled_cnn_amd import ops
  from tests.cpu_double import CpuDouble
  ops.set_library(CpuDouble(), is_double=True)
  yield
  ops.set_library(None, is_double=False)


@pytest.fixture(scope='session')
def hip_lib():
  """The real library on a real GPU; fails loudly if either is missing."""
  import torch
  from assembled_cnn_amd import lib, ops
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  ops.set_library(None, is_double=False)
  return lib.load()


@pytest.fixture(autouse=True)
def _library_tuning_follows_the_environment():
  """monkeypatch restores the environment after a test; the cached host-side switches and the library's asm_tuning struct
  follow it"""
  yield
  from assembled_cnn_amd import lib, ops
  ops._KNOBS.clear()           # the host-side switches are cached per process (ops.knob)
  if lib._lib is not None and not ops._IS_DOUBLE:
    ops.refresh_tuning()
