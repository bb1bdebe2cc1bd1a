from tensorflow import abs, cast, div, equal, greater, less_equal, logical_and, multiply, reduce_sum, to_float  # noqa: F401,A004
