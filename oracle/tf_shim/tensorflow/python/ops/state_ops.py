from tensorflow import _assign_add as assign_add  # noqa: F401
