class _Extended(object):
  pass


class _Strategy(object):
  extended = _Extended()


class _ReplicaContext(object):
  """single replica: merge_call(fn, args) == fn(strategy, *args)"""
  def merge_call(self, merge_fn, args=(), kwargs=None):
    return merge_fn(_Strategy(), *args, **(kwargs or {}))


def get_replica_context():
  return _ReplicaContext()
