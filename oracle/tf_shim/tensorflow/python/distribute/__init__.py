"""`tensorflow.python.*` names that metric/ece_metric.py imports, mapped onto the eager shim (TEST INFRASTRUCTURE ONLY)."""
