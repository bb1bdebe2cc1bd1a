from tensorflow import Tensor, _t


def remove_squeezable_dimensions(labels, predictions, expected_rank_diff=0, name=None):
  """[TF-sem] squeeze the last dimension (if it is 1) of whichever argument has one rank too many"""
  l, p = _t(labels), _t(predictions)
  diff = p.dim() - l.dim()
  if diff == expected_rank_diff + 1 and p.shape[-1] == 1:
    p = p.squeeze(-1)
  elif diff == expected_rank_diff - 1 and l.shape[-1] == 1:
    l = l.squeeze(-1)
  return Tensor(l, getattr(labels, 'dtype', None)), Tensor(p, getattr(predictions, 'dtype', None))
