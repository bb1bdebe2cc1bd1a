from tensorflow import cond  # noqa: F401
