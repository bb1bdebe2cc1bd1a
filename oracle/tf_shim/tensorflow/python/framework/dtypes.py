from tensorflow import float32, float64, int32, int64, bool  # noqa: F401,A004
