"""Host logic against golden values produced by RUNNING the reference's own plain-Python code
(tests/golden/make_reference_golden.py, executed in the build container against /root/reference with TensorFlow
replaced by an inert stand-in).  These are the only parts of the reference that can execute without TF 1.14; the
numeric kernels stay "parity unpinned" (DESIGN.md 1c)."""
import dataclasses
import json
import os

import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_pure_python.json')))


def test_per_device_batch_size_matches_reference():
  from assembled_cnn_amd import dp
  for row in GOLD['per_device_batch_size']:
    if 'error' in row:
      with pytest.raises(ValueError) as e:
        dp.per_device_batch_size(row['batch_size'], row['num_gpus'])
      assert str(e.value) == row['error']
    else:
      assert dp.per_device_batch_size(row['batch_size'], row['num_gpus']) == row['result']


def test_block_sizes_match_reference():
  from assembled_cnn_amd.model import get_block_sizes
  from oracle import assembled_oracle as O
  assert len(GOLD['get_block_sizes']) == 14
  for row in GOLD['get_block_sizes']:
    for fn in (get_block_sizes, O.get_block_sizes):
      if 'error' in row:
        with pytest.raises(ValueError):
          fn(row['resnet_size'], row['resnet_version'])
      else:
        assert list(fn(row['resnet_size'], row['resnet_version'])) == row['result']


def test_hparams_defaults_match_reference_flags():
  from assembled_cnn_amd.train import HParams
  flags = GOLD['flag_defaults']
  assert len(flags) == 57
  fields = {f.name: f for f in dataclasses.fields(HParams)}
  shared = sorted(set(fields) & set(flags))
  assert len(shared) >= 25, shared           # every model / loss / schedule flag of the hot path is mirrored by name
  hp = HParams()
  for name in shared:
    want = flags[name]['default']
    got = getattr(hp, name)
    if name == 'resnet_version':             # an enum of strings in the reference
      want = int(want)
    if isinstance(want, (int, float)) and not isinstance(want, bool):
      assert float(got) == float(want), name
    else:
      assert got == want, (name, got, want)


def test_dataset_and_preprocessing_constants_match_reference():
  from assembled_cnn_amd import input_pipeline, train
  from oracle import assembled_oracle as O, input_oracle as IO
  inet = GOLD['data_config']['ImageNet']
  assert train.IMAGENET_NUM_CLASSES == inet['num_classes'] == O.IMAGENET_NUM_CLASSES
  assert train.IMAGENET_NUM_TRAIN_IMAGES == inet['num_images']['train']
  assert inet['default_image_size'] == 224 and inet['num_channels'] == 3
  means = GOLD['preprocessing']['CHANNEL_MEANS']
  assert [float(v) for v in O.CHANNEL_MEANS] == means
  assert [round(float(v), 2) for v in IO.CHANNEL_MEANS] == means
  assert input_pipeline._RESIZE_MIN == GOLD['preprocessing']['RESIZE_MIN']
  # the kernels carry the same three literals (mean subtraction fused into the input kernels)
  here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for src in ('misc.hip', 'input.hip'):
    text = open(os.path.join(here, 'assembled_cnn_amd', 'csrc', src)).read()
    for m in means:
      assert ('%.2ff' % m) in text, (src, m)


def test_loss_scale_rule_matches_reference():
  """official/utils/flags/_performance.py:27-42 -- the rule the reference's own flags_test.py:82-96 pins."""
  from assembled_cnn_amd.train import HParams
  for row in GOLD['loss_scale']:
    hp = HParams(dtype=row['dtype'], loss_scale=row['loss_scale'])
    assert hp.get_loss_scale() == float(row['result']), row
  assert HParams(dtype='bf16').get_loss_scale() == 1.0      # bf16 (this implementation's compute type) needs none
