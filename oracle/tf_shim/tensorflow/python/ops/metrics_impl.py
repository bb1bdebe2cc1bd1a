from tensorflow import _metric_variable as metric_variable  # noqa: F401
