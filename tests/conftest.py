import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
