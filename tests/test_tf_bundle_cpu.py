"""TensorFlow checkpoint files (tensor bundle) without TensorFlow: codec known answers, round trips, corruption
handling, and the reference's WarmStart rule through real files (utils/hook_utils.py:29-56)."""
import os
import struct

import numpy as np
import pytest
import torch

from assembled_cnn_amd import tf_bundle as B


def test_crc32c_and_masking_known_answers():
  # RFC 3720 B.4 / LevelDB util/crc32c_test.cc
  assert B.crc32c(b'\x00' * 32) == 0x8a9136aa
  assert B.crc32c(b'\xff' * 32) == 0x62a8ab43
  assert B.crc32c(bytes(range(32))) == 0x46dd794e
  assert B.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
  assert B.crc32c(b'123456789') == 0xe3069283
  assert B.crc32c(b'world', B.crc32c(b'hello ')) == B.crc32c(b'hello world')      # Extend
  c = B.crc32c(b'foo')
  assert B.mask_crc(c) != c and B.unmask_crc(B.mask_crc(c)) == c and B.mask_crc(B.mask_crc(c)) != c
  assert B.unmask_crc(B.unmask_crc(B.mask_crc(B.mask_crc(c)))) == c


def test_varint_and_entry_codec():
  for n in (0, 1, 127, 128, 300, 2 ** 32 - 1, 2 ** 40 + 5):
    enc = B._put_varint(n)
    assert B._get_varint(enc, 0) == (n, len(enc))
  assert B._put_varint(300) == b'\xac\x02'                                         # protobuf documentation example
  e = B._parse_entry(B._encode_entry(1, (3, 3, 64, 128), 4096, 294912, 0xdeadbeef))
  assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c']) == (1, (3, 3, 64, 128), 4096, 294912, 0xdeadbeef)
  assert B._parse_entry(B._encode_entry(9, (), 0, 8, 1))['shape'] == ()          # scalar (global_step)


def test_bundle_round_trip_many_blocks_and_dtypes(tmp_path):
  rng = np.random.default_rng(0)
  names = ['resnet_model/conv2d_%d/kernel' % i for i in range(150)] + ['resnet_model/dense/bias', 'global_step']
  vs = {n: rng.normal(size=(1, 1, 4 + i % 3, 5)).astype(np.float32) for i, n in enumerate(names[:150])}
  vs['resnet_model/dense/bias'] = rng.normal(size=(1001,)).astype(np.float64)
  vs['global_step'] = np.asarray(123456789012, dtype=np.int64)
  vs['flags'] = np.array([True, False, True])
  vs['half'] = rng.normal(size=(7,)).astype(np.float16)
  prefix = str(tmp_path / 'model.ckpt-5')
  B.write_bundle(prefix, vs, block_size=512)       # forces a multi-block index with prefix compression
  idx = B.read_index(prefix + '.index')
  assert idx['']['num_shards'] == 1 and list(idx)[1:] == sorted(vs)
  back = B.read_bundle(prefix, verify_data=True)
  assert list(back) == sorted(vs)
  for n, a in vs.items():
    assert back[n].dtype == a.dtype and back[n].shape == a.shape and np.array_equal(back[n], a), n
  assert int(back['global_step']) == 123456789012
  only = B.read_bundle(prefix, names=['flags', 'half'])
  assert list(only) == ['flags', 'half']
  with pytest.raises(KeyError):
    B.read_bundle(prefix, names=['nope'])
  # footer: magic number and 48 bytes
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
  # a flipped byte in a block fails its checksum; a flipped byte in the data fails verify_data
  bad = bytearray(raw)
  bad[10] ^= 0x40
  open(prefix + '.index', 'wb').write(bytes(bad))
  with pytest.raises(ValueError):
    B.read_index(prefix + '.index')
  open(prefix + '.index', 'wb').write(raw)
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[3] ^= 1
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError):
    B.read_bundle(prefix, verify_data=True)
  with pytest.raises(ValueError):
    open(str(tmp_path / 'junk.index'), 'wb').write(b'x' * 100)
    B.read_index(str(tmp_path / 'junk.index'))


def test_bfloat16_variables_widen_to_float32(tmp_path):
  prefix = str(tmp_path / 'bf')
  a = np.array([1.0, -2.5, 3.140625], dtype=np.float32)
  B.write_bundle(prefix, {'v': a})
  # rewrite the entry as DT_BFLOAT16 over the upper halves of the floats
  hi = (a.view(np.uint32) >> 16).astype('<u2').tobytes()
  open(prefix + '.data-00000-of-00001', 'wb').write(hi)
  out = bytearray()
  blk = B._BlockBuilder()
  blk.add(b'', B._pb_varint_field(1, 1))
  blk.add(b'v', B._encode_entry(B.DT_BFLOAT16, (3,), 0, len(hi), B.mask_crc(B.crc32c(hi))))
  index = B._BlockBuilder(1)
  index.add(b'v', B._emit_block(out, blk.finish()))
  meta = B._emit_block(out, B._BlockBuilder().finish())
  ih = B._emit_block(out, index.finish())
  out += (meta + ih).ljust(40, b'\x00') + struct.pack('<Q', B.TABLE_MAGIC)
  open(prefix + '.index', 'wb').write(bytes(out))
  assert np.array_equal(B.read_bundle(prefix, verify_data=True)['v'], a)


def test_model_round_trip_and_warm_start_through_tf_checkpoint_files(cpu_double, tmp_path):
  """export -> TF checkpoint files -> a second model: full restore (weights, moving statistics, Momentum slots) and the
  WarmStartHook rule (classifier skipped, only at global_step 0, accumulators untouched); TF layouts on disk."""
  from assembled_cnn_amd import checkpoint as ck
  from assembled_cnn_amd.model import Model
  kw = dict(resnet_size=50, num_classes=1001, device='cpu', use_se_block=True)
  a = Model(seed=1, **kw)
  a.build((64, 64))
  a.arena.m32.normal_(0, 0.01)
  a.arena.state.uniform_(0.5, 1.5)
  prefix = str(tmp_path / 'model.ckpt-77')
  ck.save_tf_checkpoint(prefix, a, global_step=77)
  disk = ck.load_tf_checkpoint(str(tmp_path))           # through the `checkpoint` state file, like latest_checkpoint
  assert disk['resnet_model/conv2d/kernel'].shape == (7, 7, 3, 64)                 # HWIO
  assert disk['resnet_model/dense/kernel'].shape == (2048, 1001)                   # [in, units]
  assert disk['resnet_model/dense/kernel/Momentum'].shape == (2048, 1001)
  assert disk['resnet_model/se_block/seblock_dense_1/kernel'].shape == (1, 1, 256, 16)
  assert int(disk['global_step']) == 77
  b = Model(seed=2, **kw)
  b.build((64, 64))
  rep = ck.import_variables(b, disk)
  assert not rep['missing'] and not rep['missing_slots']
  for n in a.arena.specs:       # (the flat arenas pad every variable to 8 elements: compare the variables, not the padding)
    assert torch.equal(a.arena.w(n), b.arena.w(n)) and torch.equal(a.arena.m(n), b.arena.m(n)), n
  for n in a.arena.state_specs:
    assert torch.equal(a.arena.st(n), b.arena.st(n)), n
  # warm start into a used model: classifier kept, accumulators kept
  c = Model(seed=3, **kw)
  c.build((64, 64))
  c.arena.m32.fill_(0.5)
  dense_before = c.arena.w('resnet_model/dense/kernel').clone()
  rep = ck.import_variables(c, disk, warm_start=True, global_step=0)
  assert 'resnet_model/dense/kernel' in rep['skipped'] and 'resnet_model/se_block/seblock_dense_1/kernel' in rep['loaded']
  assert torch.equal(c.arena.w('resnet_model/dense/kernel'), dense_before)
  assert torch.equal(c.arena.w('resnet_model/conv2d/kernel'), a.arena.w('resnet_model/conv2d/kernel'))
  assert float(c.arena.m('resnet_model/conv2d/kernel').min()) == 0.5
  assert ck.import_variables(c, disk, warm_start=True, global_step=5)['loaded'] == []
  # a slot-less (inference) checkpoint restored in full: accumulators are zeroed and reported
  slotless = {k: v for k, v in disk.items() if not k.endswith('/Momentum')}
  rep = ck.import_variables(c, slotless)
  assert len(rep['missing_slots']) == len(c.arena.specs) and all(float(c.arena.m(n).abs().max()) == 0.0 for n in c.arena.specs)
  assert ck._is_dense_kernel('resnet_model/dense/kernel/Momentum') and not ck._is_dense_kernel('kernel/Momentum')
  assert not ck._is_dense_kernel('resnet_model/embedding_dense/kernel')


def test_lane_parallel_crc32c_equals_the_bytewise_form():
  """long buffers take the lane-parallel numpy path (CRC is linear over GF(2)); it must agree with the table walk at every
  length around the lane switch, for odd lengths (zero padding in front) and through the Extend form"""
  rng = np.random.default_rng(0)
  for n in (65535, 65536, 65537, 100003, (1 << 20) + 7):
    d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    ref = B._crc32c_bytes(d)
    assert B.crc32c(d) == ref
    k = n // 3
    assert B.crc32c(d[k:], B.crc32c(d[:k])) == ref
