from tensorflow import variable_scope  # noqa: F401
