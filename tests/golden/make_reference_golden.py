#!/usr/bin/env python
"""Generate tests/golden/reference_pure_python.json by RUNNING the reference's own code (run in the build container,
where /root/reference exists; the GPU box never needs it).

TensorFlow 1.14 is not installable here, so only the reference functions that are plain Python can be executed:
their modules are imported with `tensorflow` replaced by an inert stand-in, and the functions below never touch it.
  official/utils/misc/distribution_utils.py:48-76   per_device_batch_size  (+ its ValueError text)
  functions/model_fns.py:98-135                     get_block_sizes        (+ ValueError on unknown sizes)
  functions/data_config.py                          dataset constants used by the schedule / class count
  nets/hparams_config.py                            every flag's type and default (absl.flags replaced by a recorder)
  preprocessing/imagenet_preprocessing.py:46-54     CHANNEL_MEANS, _RESIZE_MIN
  official/utils/flags/_performance.py:27-42        get_loss_scale
"""
import json
import os
import sys
import types
from unittest import mock

REF = '/root/reference'


def main():
  sys.path.insert(0, REF)
  # every `tensorflow[.x.y]`, `absl[...]`, `tensorflow_hub` import resolves to one inert stand-in package
  import importlib.abc
  import importlib.machinery
  fake = mock.MagicMock(name='tensorflow')
  fake.VERSION = fake.__version__ = '1.14.0'      # README.md:85; read by metric/ece_metric.py:19-21 at import time
  fake.__path__ = []

  class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
      return fake

    def exec_module(self, module):
      pass

  class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
      if name.split('.')[0] in ('tensorflow', 'absl', 'tensorflow_hub'):
        return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
      return None
  sys.meta_path.insert(0, _Finder())
  from official.utils.misc import distribution_utils as du
  out = {'per_device_batch_size': [], 'get_block_sizes': [], 'data_config': {}}
  for bs, n in [(256, 1), (256, 0), (2048, 8), (1024, 8), (256, 2), (256, 4), (100, 8), (7, 3), (1023, 8), (48, 6)]:
    try:
      out['per_device_batch_size'].append(dict(batch_size=bs, num_gpus=n, result=du.per_device_batch_size(bs, n)))
    except ValueError as e:
      out['per_device_batch_size'].append(dict(batch_size=bs, num_gpus=n, error=str(e)))
  try:
    from functions import model_fns
    for size in (18, 34, 50, 101, 152, 200, 77):
      for ver in (1, 2):
        try:
          out['get_block_sizes'].append(dict(resnet_size=size, resnet_version=ver,
                                             result=list(model_fns.get_block_sizes(size, ver))))
        except ValueError:
          out['get_block_sizes'].append(dict(resnet_size=size, resnet_version=ver, error='ValueError'))
  except Exception as e:   # pragma: no cover
    out['get_block_sizes_import_error'] = repr(e)
  try:
    from functions import data_config as dc
    for cls in ('Default', 'ImageNet'):
      c = getattr(dc, cls)
      out['data_config'][cls] = {k: getattr(c, k) for k in dir(c)
                                 if not k.startswith('_') and isinstance(getattr(c, k), (int, float, str, dict))}
  except Exception as e:   # pragma: no cover
    out['data_config_import_error'] = repr(e)
  # flag defaults: call the reference's define_* functions with a recorder in place of absl.flags
  try:
    from nets import hparams_config as hc

    class _Rec(object):
      def __init__(self):
        self.flags = {}

      def __getattr__(self, kind):
        if not kind.startswith('DEFINE_'):
          return mock.MagicMock()

        def define(*a, **kw):
          name = kw.get('name', a[0] if a else None)
          default = kw.get('default', a[1] if len(a) > 1 else None)
          self.flags[name] = dict(type=kind[len('DEFINE_'):], default=default)
        return define
    rec = _Rec()
    for fn in dir(hc):
      if fn.startswith('define_') and callable(getattr(hc, fn)):
        try:
          getattr(hc, fn)(rec)
        except TypeError:
          pass
    out['flag_defaults'] = rec.flags
  except Exception as e:   # pragma: no cover
    out['flag_defaults_import_error'] = repr(e)
  try:   # official/utils/flags/_performance.py:27-42 (pinned by the reference's own flags_test.py:82-96)
    from official.utils.flags import _performance as perf
    ns = types.SimpleNamespace
    out['loss_scale'] = [dict(dtype=dt, loss_scale=ls, result=perf.get_loss_scale(ns(dtype=dt, loss_scale=ls)))
                         for dt in ('fp16', 'fp32') for ls in (None, 1, 64, 1024)]
  except Exception as e:   # pragma: no cover
    out['loss_scale_import_error'] = repr(e)
  try:
    from preprocessing import imagenet_preprocessing as ip
    out['preprocessing'] = dict(CHANNEL_MEANS=list(ip.CHANNEL_MEANS), RESIZE_MIN=ip._RESIZE_MIN)
  except Exception as e:   # pragma: no cover
    out['preprocessing_import_error'] = repr(e)
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_pure_python.json')
  json.dump(out, open(path, 'w'), indent=1, sort_keys=True, default=str)
  print('wrote', path, {k: (len(v) if hasattr(v, '__len__') else v) for k, v in out.items()})


if __name__ == '__main__':
  main()
