"""Reader / writer for TensorFlow's checkpoint format ("tensor bundle", ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``)
without TensorFlow -- the on-disk format of the reference's checkpoints.

The reference restores its published Assemble-ResNet checkpoints through ``tf.train.Saver`` (utils/hook_utils.py:29-56 for
warm starts, the Estimator for evaluation, README.md:158-164).  With this module a user holding such a checkpoint can load
it straight into :mod:`assembled_cnn_amd.checkpoint` (``read_bundle`` -> ``import_variables``) and write the trained
variables back in the same format (``write_bundle``), so the two code bases exchange weights through the reference's own
files.

Format, restated from TensorFlow 1.14's published sources (no TensorFlow file exists on this box to pin it against; the
CRC-32C and varint codecs are pinned by the RFC 3720 / LevelDB test vectors in tests/test_tf_bundle_cpu.py, the rest by
round trips):

  * ``<prefix>.index`` is a LevelDB-style sorted string table (tensorflow/core/lib/io/table, format.cc): data blocks, a
    meta-index block, an index block and a 48-byte footer (two BlockHandles = varint64 offset + size, padding, the magic
    number 0xdb4775248b80fb57).  A block is a run of prefix-compressed entries -- varint32 shared, varint32 non_shared,
    varint32 value_length, key suffix, value -- followed by the uint32 restart offsets and their count; in the file every
    block is followed by a 1-byte compression type (0 = none; the bundle writer never compresses) and a masked CRC-32C.
  * key "" holds a BundleHeaderProto (num_shards = 1, endianness (2) = LITTLE = 0, version (3)); every other key is a variable name
    whose value is a BundleEntryProto: dtype (1), shape (2: TensorShapeProto, repeated dim {size = 1}), shard_id (3),
    offset (4), size (5), crc32c (6, fixed32, masked) (tensorflow/core/protobuf/tensor_bundle.proto).
  * the data shard holds the raw little-endian row-major tensor bytes at [offset, offset + size).
"""
from __future__ import annotations

import os
import re
import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'), 6: np.dtype('i1'),
      9: np.dtype('<i8'), 10: np.dtype('?'), 19: np.dtype('<f2')}
DT_BFLOAT16 = 14
_DT_OF = {v: k for k, v in DT.items()}


# ---- CRC-32C (Castagnoli), LevelDB masking --------------------------------------------------------------------------
def _make_table():
  tab = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
    tab.append(c)
  return tab


_CRC_TABLE = _make_table()
_CRC_NP = np.array(_CRC_TABLE, dtype=np.uint32)


def _crc32c_bytes(data, crc: int = 0) -> int:
  """the reference byte-at-a-time form (~6 MB/s in CPython): short inputs and the check of the lane-parallel form"""
  c = crc ^ 0xffffffff
  tab = _CRC_TABLE
  for b in data:
    c = tab[(c ^ b) & 0xff] ^ (c >> 8)
  return c ^ 0xffffffff


# The CRC register update is linear over GF(2) in (register, data): the register after n more bytes is
# M_n . register  ^  raw(data), with M_n the 32 x 32 matrix of "feed n zero bytes" and raw() the register run from 0.
# That makes a long buffer lane-parallel: cut it into L equal lanes (leading zero padding is free for a zero register),
# run all lanes through the table at once with numpy (m = n / L vector steps instead of n Python steps), then fold the
# lane registers pairwise with M_m, M_2m, ... (log2 L vector steps).  ~70 MB/s instead of 6: a 200 MB checkpoint is
# checksummed in about three seconds, so read_bundle verifies the data by default (as TensorFlow's BundleReader does).
def _zero_byte_matrix():
  return [(_CRC_TABLE[(1 << i) & 0xff] ^ ((1 << i) >> 8)) & 0xffffffff for i in range(32)]


def _mat_vec(mat, v: int) -> int:
  out, i = 0, 0
  while v:
    if v & 1:
      out ^= mat[i]
    v >>= 1
    i += 1
  return out


def _mat_mul(a, b):
  """(a . b): apply b first"""
  return [_mat_vec(a, col) for col in b]


def _shift_matrix(nbytes: int):
  """matrix of feeding ``nbytes`` zero bytes"""
  result = [1 << i for i in range(32)]
  base = _zero_byte_matrix()
  while nbytes:
    if nbytes & 1:
      result = _mat_mul(base, result)
    base = _mat_mul(base, base)
    nbytes >>= 1
  return result


def _mat_vec_np(mat, v: np.ndarray) -> np.ndarray:
  out = np.zeros_like(v)
  for i in range(32):
    out ^= np.where((v >> np.uint32(i)) & np.uint32(1), np.uint32(mat[i]), np.uint32(0)).astype(np.uint32)
  return out


_CRC_LANE_MIN = 1 << 16


def crc32c(data, crc: int = 0) -> int:
  """CRC-32C (Castagnoli) of ``data`` (bytes-like), continuing from ``crc`` (the Extend form of LevelDB / TensorFlow)"""
  buf = np.frombuffer(memoryview(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
  n = int(buf.size)
  if n < _CRC_LANE_MIN:
    return _crc32c_bytes(buf.tobytes(), crc)
  lanes = 1
  while lanes < 16384 and n // (lanes * 2) >= 2048:
    lanes *= 2
  m = -(-n // lanes)
  pad = lanes * m - n
  if pad:
    buf = np.concatenate([np.zeros(pad, dtype=np.uint8), buf])
  cols = np.ascontiguousarray(buf.reshape(lanes, m).T)          # step j touches row j: contiguous
  c = np.zeros(lanes, dtype=np.uint32)
  tab = _CRC_NP
  for j in range(m):
    c = tab[(c ^ cols[j]) & np.uint32(0xff)] ^ (c >> np.uint32(8))
  span = m
  while c.size > 1:                                              # fold neighbours: left . x^(8 span) + right
    c = _mat_vec_np(_shift_matrix(span), c[0::2]) ^ c[1::2]
    span *= 2
  raw = int(c[0])
  return (raw ^ _mat_vec(_shift_matrix(n), (crc ^ 0xffffffff) & 0xffffffff)) ^ 0xffffffff


def mask_crc(crc: int) -> int:
  return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked: int) -> int:
  rot = (masked - _MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / minimal protobuf -------------------------------------------------------------------------------------
def _put_varint(n: int) -> bytes:
  out = bytearray()
  while n >= 0x80:
    out.append((n & 0x7f) | 0x80)
    n >>= 7
  out.append(n)
  return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  shift = n = 0
  while True:
    b = buf[pos]
    pos += 1
    n |= (b & 0x7f) << shift
    if not b & 0x80:
      return n, pos
    shift += 7
    if shift > 63:
      raise ValueError('malformed varint')


def _pb_fields(buf: bytes):
  """(field number, wire type, value) of a serialised message; value is an int (varint / fixed) or bytes"""
  pos = 0
  while pos < len(buf):
    key, pos = _get_varint(buf, pos)
    fn, wt = key >> 3, key & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wt == 2:
      n, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + n])
      pos += n
    elif wt == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wt)
    yield fn, wt, v


def _pb_varint_field(fn: int, v: int) -> bytes:
  return _put_varint(fn << 3) + _put_varint(v) if v else b''


def _pb_bytes_field(fn: int, v: bytes) -> bytes:
  return _put_varint((fn << 3) | 2) + _put_varint(len(v)) + v


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
  dims = []
  for fn, _, v in _pb_fields(buf):
    if fn == 2:                      # repeated Dim dim = 2
      size = 0
      for f2, _, v2 in _pb_fields(v):
        if f2 == 1:
          size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
      dims.append(size)
    elif fn == 3 and v:
      raise ValueError('tensor of unknown rank in a checkpoint')
  return tuple(dims)


def _parse_entry(buf: bytes) -> dict:
  e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=0, sliced=False)
  for fn, _, v in _pb_fields(buf):
    if fn == 1:
      e['dtype'] = v
    elif fn == 2:
      e['shape'] = _parse_shape(v)
    elif fn == 3:
      e['shard_id'] = v
    elif fn == 4:
      e['offset'] = v
    elif fn == 5:
      e['size'] = v
    elif fn == 6:
      e['crc32c'] = v
    elif fn == 7:
      e['sliced'] = True
  return e


def _encode_entry(dtype: int, shape, offset: int, size: int, crc: int) -> bytes:
  dims = b''.join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
  return (_pb_varint_field(1, dtype) + _pb_bytes_field(2, dims) + _pb_varint_field(3, 0) + _pb_varint_field(4, offset) +
          _pb_varint_field(5, size) + _put_varint((6 << 3) | 5) + struct.pack('<I', crc))


# ---- table (read) ---------------------------------------------------------------------------------------------------
def _read_block(f: bytes, offset: int, size: int) -> bytes:
  raw = f[offset:offset + size]
  if len(raw) != size or offset + size + 5 > len(f):
    raise ValueError('truncated table block')
  ctype = f[offset + size]
  stored = struct.unpack_from('<I', f, offset + size + 1)[0]
  if unmask_crc(stored) != crc32c(f[offset:offset + size + 1]):
    raise ValueError('table block checksum mismatch (corrupt .index file)')
  if ctype == 1:
    raise NotImplementedError('snappy-compressed table block (TensorFlow writes checkpoint indices uncompressed)')
  if ctype != 0:
    raise ValueError('unknown block compression type %d' % ctype)
  return raw


def _block_entries(block: bytes):
  if len(block) < 4:
    raise ValueError('bad table block')
  num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * num_restarts
  if limit < 0:
    raise ValueError('bad restart array')
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key):
      raise ValueError('bad key prefix')
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_index(path: str) -> "OrderedDict[str, dict]":
  """name -> {dtype, shape, shard_id, offset, size, crc32c}; '' -> the header {'num_shards': ...}"""
  with open(path, 'rb') as fh:
    f = fh.read()
  if len(f) < 48:
    raise ValueError('%s is too short to be a checkpoint index' % path)
  footer = f[-48:]
  if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
    raise ValueError('%s is not a TensorFlow checkpoint index (bad magic number)' % path)
  pos = 0
  _, pos = _get_varint(footer, pos)      # metaindex handle
  _, pos = _get_varint(footer, pos)
  ioff, pos = _get_varint(footer, pos)   # index handle
  isize, pos = _get_varint(footer, pos)
  out: "OrderedDict[str, dict]" = OrderedDict()
  for _, handle in _block_entries(_read_block(f, ioff, isize)):
    boff, p2 = _get_varint(handle, 0)
    bsize, _ = _get_varint(handle, p2)
    for key, value in _block_entries(_read_block(f, boff, bsize)):
      if key == b'':
        hdr = dict(num_shards=0, endianness=0)
        for fn, _, v in _pb_fields(value):
          if fn == 1:
            hdr['num_shards'] = v
          elif fn == 2:
            hdr['endianness'] = v
        if hdr['endianness'] != 0:
          raise NotImplementedError('big-endian checkpoint')
        out[''] = hdr
      else:
        out[key.decode('utf-8')] = _parse_entry(value)
  return out


def _shard_name(prefix: str, shard: int, num: int) -> str:
  return '%s.data-%05d-of-%05d' % (prefix, shard, num)


def read_bundle(prefix: str, names=None, verify_data: bool = True) -> "OrderedDict[str, np.ndarray]":
  """Every (or the named) tensor of the checkpoint ``prefix`` as numpy arrays (bfloat16 widened to float32)."""
  idx = read_index(prefix + '.index')
  num = idx.get('', {}).get('num_shards', 1) or 1
  out: "OrderedDict[str, np.ndarray]" = OrderedDict()
  handles = {}
  try:
    for name, e in idx.items():
      if name == '' or (names is not None and name not in names):
        continue
      if e['sliced']:
        raise NotImplementedError('variable %s is stored in slices (partitioned variable)' % name)
      fh = handles.get(e['shard_id'])
      if fh is None:
        fh = handles[e['shard_id']] = open(_shard_name(prefix, e['shard_id'], num), 'rb')
      fh.seek(e['offset'])
      raw = fh.read(e['size'])
      if len(raw) != e['size']:
        raise ValueError('variable %s: data shard is truncated' % name)
      if verify_data and unmask_crc(e['crc32c']) != crc32c(raw):
        raise ValueError('variable %s: checksum mismatch' % name)
      if e['dtype'] == DT_BFLOAT16:
        a = (np.frombuffer(raw, dtype='<u2').astype(np.uint32) << 16).view(np.float32)
      elif e['dtype'] in DT:
        a = np.frombuffer(raw, dtype=DT[e['dtype']])
      else:
        raise NotImplementedError('variable %s has TensorFlow dtype %d' % (name, e['dtype']))
      n = int(np.prod(e['shape'])) if e['shape'] else 1
      if a.size != n:
        raise ValueError('variable %s: %d elements on disk, shape %s' % (name, a.size, e['shape']))
      out[name] = a.reshape(e['shape']).copy()
  finally:
    for fh in handles.values():
      fh.close()
  if names is not None:
    missing = [n for n in names if n not in out]
    if missing:
      raise KeyError('variables missing from the checkpoint: %s' % missing[:5])
  return out


# ---- table (write) --------------------------------------------------------------------------------------------------
class _BlockBuilder(object):
  def __init__(self, restart_interval=16):
    self.buf = bytearray()
    self.restarts = [0]
    self.count = 0
    self.last = b''
    self.interval = restart_interval

  def add(self, key: bytes, value: bytes):
    shared = 0
    if self.count < self.interval:
      while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.count = 0
    self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
    self.last = key
    self.count += 1

  def finish(self) -> bytes:
    return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _emit_block(out: bytearray, block: bytes) -> bytes:
  handle = _put_varint(len(out)) + _put_varint(len(block))
  out += block + b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00')))
  return handle


def write_bundle(prefix: str, variables: Dict[str, np.ndarray], block_size: int = 4096):
  """``variables`` (name -> array: float32 / float64 / int32 / int64 / uint8 / bool / float16) as a one-shard checkpoint.
  Keys are written in byte order, data 'offset'-packed in the same order, as tf.train.Saver does."""
  items = sorted(((k.encode('utf-8'), np.asarray(v)) for k, v in variables.items()), key=lambda kv: kv[0])
  entries = []
  with open(_shard_name(prefix, 0, 1), 'wb') as fh:
    off = 0
    for key, a in items:
      dt = a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype
      code = _DT_OF.get(np.dtype(dt))
      if code is None:
        raise NotImplementedError('dtype %s of variable %s' % (a.dtype, key.decode()))
      raw = a.astype(dt, copy=False).tobytes(order='C')
      fh.write(raw)
      entries.append((key, _encode_entry(code, a.shape, off, len(raw), mask_crc(crc32c(raw)))))
      off += len(raw)
  header = _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))   # num_shards = 1, version {producer: 1}
  out = bytearray()
  index = _BlockBuilder(restart_interval=1)
  blk = _BlockBuilder()
  last_key = b''
  for key, value in [(b'', header)] + entries:
    blk.add(key, value)
    last_key = key
    if len(blk.buf) >= block_size:
      index.add(last_key, _emit_block(out, blk.finish()))
      blk = _BlockBuilder()
  if blk.buf:
    index.add(last_key, _emit_block(out, blk.finish()))
  meta = _emit_block(out, _BlockBuilder().finish())
  ih = _emit_block(out, index.finish())
  footer = meta + ih
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out += footer
  with open(prefix + '.index', 'wb') as fh:
    fh.write(bytes(out))


def write_checkpoint_state(directory: str, prefix_basename: str):
  """the ``checkpoint`` text file tf.train.latest_checkpoint reads"""
  with open(os.path.join(directory, 'checkpoint'), 'w') as fh:
    fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (prefix_basename, prefix_basename))


def latest_checkpoint(directory: str) -> Optional[str]:
  """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` in ``<directory>/checkpoint``."""
  state = os.path.join(directory, 'checkpoint')
  if not os.path.exists(state):
    return None
  with open(state) as fh:
    m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', fh.read(), re.M)
  if not m:
    return None
  p = m.group(1)
  p = p if os.path.isabs(p) else os.path.join(directory, p)
  return p if os.path.exists(p + '.index') else None
