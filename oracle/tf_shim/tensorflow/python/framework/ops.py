import contextlib

from tensorflow import convert_to_tensor  # noqa: F401


def add_to_collections(names, value):
  return None


@contextlib.contextmanager
def control_dependencies(control_inputs):
  yield
