"""Data-parallel path on CPU: world_size 2 over gloo, host code driven through the C-ABI test double.

Invariants: after a DP step every replica holds identical weights; the exchanged gradient
is the SUM over replicas of the per-shard gradients (BN statistics stay per replica), divided by the
number of replicas inside the optimiser; buckets are launched by the backward watermark."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


class _FakeStream(object):
  def __init__(self, name):
    self.name = name

  def wait_stream(self, other):
    pass


def _two_stream_tape():
  """Make the walker take its GPU path on the CPU double: the big branch of every BigLittle stage goes through
  model._bl_backward (interleaved with the little branch, notifications deferred), with stand-in streams."""
  import contextlib
  from assembled_cnn_amd import model as pmodel
  main, side, cur = _FakeStream('main'), _FakeStream('side'), []
  cur.append(main)

  @contextlib.contextmanager
  def ctx(s):
    cur.append(s)
    try:
      yield
    finally:
      cur.pop()
  pmodel._current_stream = lambda: cur[-1]
  pmodel._stream_ctx = ctx
  pmodel.Model._branch_stream = lambda self, c, x: None if c.dry else side


def _worker(rank, world, port, out_dir, two_stream_tape=False):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  torch.set_num_threads(2)
  if two_stream_tape:
    _two_stream_tape()
  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  from tests.cpu_double import CpuDouble
  from tests import model_parity as mpar
  ops.set_library(CpuDouble(), is_double=True)
  dp.init_process_group_from_env('gloo')
  hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
               zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01, weight_decay=1e-4,
               batch_size=2 * world)
  B, S = 2, 64
  img, _, labels = mpar.inputs(B * world, S)

  # reference: per-shard gradients computed locally, without any exchange
  ref = Trainer(hp, seed=0, device='cpu')
  ref.model.build((S, S))
  shard_grads = []
  for r in range(world):
    x, onehot, _ = ref.prepare_inputs(img[r * B:(r + 1) * B], labels[r * B:(r + 1) * B])
    ref.model(x, True, prepadded=True)
    rows, dz = ops.softmax_ce(ref.model.logits_padded, ref.model.ldc, onehot, None, B, 1001, 0.0, 0.0, 1.0,
                              ref.model.ldc)
    ref.model.backward(dz)
    shard_grads.append(ref.model.arena.g32.clone())
  w0 = ref.model.arena.w32.clone()

  tr = Trainer(hp, seed=0, device='cpu', world_size=world)
  tr.model.build((S, S))
  sync = dp.GradSync(tr.model.arena, bucket_bytes=8 << 20)
  launched = []
  orig = sync._launch
  sync._launch = lambda s, i: (launched.append((s, i, len(tr.model._ctx.tape) if tr.model._ctx else -1)), orig(s, i))[1]
  tr.grad_sync = sync
  assert torch.equal(tr.model.arena.w32, w0)
  tr.train_step(img[rank * B:(rank + 1) * B], labels[rank * B:(rank + 1) * B])

  g_sum = shard_grads[0] + shard_grads[1]
  assert torch.equal(tr.model.arena.g32, g_sum), 'exchanged gradient must be the SUM of the shard gradients'
  assert len(launched) == sum(len(b) for b in sync.segments) and len(sync.segments[0]) >= 4
  # identical weights on every replica
  gathered = [torch.empty_like(tr.model.arena.w32) for _ in range(world)]
  dist.all_gather(gathered, tr.model.arena.w32)
  assert torch.equal(gathered[0], gathered[1])
  # and they equal one momentum-SGD step on the mean gradient
  nd = tr.model.arena.decay_elems
  g = g_sum / world
  exp = w0.clone()
  exp[:nd] -= 0.01 * (g[:nd] + 1e-4 * w0[:nd])
  exp[nd:] -= 0.01 * g[nd:]
  assert torch.allclose(tr.model.arena.w32, exp, rtol=1e-5, atol=1e-7)
  # the recorded step's bookkeeping (Trainer.capture with the exchange attached): while the backward pass is being
  # RECORDED the bucket launches only cut the tape and are written down, in the eager step's order; replayed one by one
  # behind their segments they exchange the same buckets, and finish() hands over the rest: the same summed gradient
  rec = Trainer(hp, seed=0, device='cpu', world_size=world)
  rec.model.build((S, S))
  rsync = dp.GradSync(rec.model.arena, bucket_bytes=8 << 20)
  rec.grad_sync = rsync
  tape = ops.tape_begin()
  rsync.begin_recording()
  rec._forward_backward(img[rank * B:(rank + 1) * B], labels[rank * B:(rank + 1) * B], None, None)
  rsync.end_recording()
  assert ops.tape_end() == tape and ops.tape_info(tape)['segments'] == len(rsync.recorded) + 1
  assert rsync.recorded == [(s_, i_) for (s_, i_, _) in launched], 'recorded bucket order == the eager step\'s launch order'
  assert torch.equal(rec.model.arena.g32, shard_grads[rank]), 'nothing is exchanged while recording'
  for k in range(len(rsync.recorded)):
    rsync.launch_recorded(k)
  rsync.finish()
  assert torch.equal(rec.model.arena.g32, g_sum)
  late = next(k for k, (_, i_) in enumerate(rsync.recorded) if i_ > 0)     # a segment's second bucket, before its first:
  with pytest.raises(RuntimeError):                                         # refused (no collective is issued)
    rsync.launch_recorded(late)
  rec.model.arena.on_grad = None
  # evaluation metrics: each rank scores its own shard, eval_result() all-reduces the 33 running sums
  # (metric/ece_metric.py:281-298) and equals a single-process evaluation of the whole batch
  from oracle import assembled_oracle as O
  xe = img.float() - torch.tensor(O.CHANNEL_MEANS)
  tr.eval_reset()
  tr.eval_step(xe[rank * B:(rank + 1) * B], labels[rank * B:(rank + 1) * B])
  local = tr.eval_result(reduce=False)
  both = tr.eval_result()
  assert local['count'] == B and both['count'] == B * world
  single = Trainer(hp, seed=0, device='cpu')
  single.model.build((S, S))
  single.model.arena.w32.copy_(tr.model.arena.w32)
  single.model.arena.state.copy_(tr.model.arena.state)
  single.model.arena.refresh_shadows()
  single.eval_reset()
  single.eval_step(xe, labels)
  want = single.eval_result(reduce=False)
  for k in ('accuracy', 'accuracy_top_5', 'count'):
    assert abs(both[k] - want[k]) < 1e-6, (k, both, want)
  # confidences move by bf16 rounding when the CPU double convolves a batch of 4 instead of 2 x 2
  assert abs(both['ece'] - want['ece']) < 1e-4, (both, want)
  # the reduction itself is exact: reduced sums == sum of the per-rank sums
  gathered_st = [torch.empty_like(tr.eval_state) for _ in range(world)]
  dist.all_gather(gathered_st, tr.eval_state)
  st_sum = gathered_st[0] + gathered_st[1]
  n = float(st_sum[2])
  assert both['accuracy'] == float(st_sum[0]) / n and both['accuracy_top_5'] == float(st_sum[1]) / n
  if rank == 0:
    open(os.path.join(out_dir, 'ok'), 'w').write('%d buckets' % len(launched))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('two_stream_tape', [False, True], ids=['tape-order', 'biglittle-interleaved'])
def test_dp_world2_gloo(tmp_path, two_stream_tape):
  """second run: the tape order of the GPU (big branch of each BigLittle stage interleaved with the little branch, its
  gradient-ready notifications deferred) under the same invariants -- real bucket launches over gloo while the tape runs"""
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path), two_stream_tape), nprocs=2, join=True)
  assert (tmp_path / 'ok').exists()


def test_per_device_batch_size():
  """official/utils/misc/distribution_utils_test.py:54-64."""
  from assembled_cnn_amd.dp import per_device_batch_size
  assert per_device_batch_size(147, 7) == 21
  assert per_device_batch_size(32, 1) == 32 and per_device_batch_size(32, 0) == 32
  with pytest.raises(ValueError):
    per_device_batch_size(147, 5)


def _worker_mixup(rank, world, port, out_dir, comm_dtype):
  """SURVEY 8e invariant at world size 4 with mixup type 1: every rank receives its own 2B-image slice, forms the
  mixup pairs INSIDE that slice (functions/input_fns.py:98-104), normalises with its own batch statistics, and the
  exchanged gradient is the sum of the four per-shard gradients."""
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  torch.set_num_threads(2)
  import numpy as np
  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  from tests.cpu_double import CpuDouble
  from tests import model_parity as mpar
  ops.set_library(CpuDouble(), is_double=True)
  dp.init_process_group_from_env('gloo')
  B, S = 2, 32
  hp = HParams(resnet_version=1, zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01,
               weight_decay=1e-4, label_smoothing=0.1, mixup_type=1, batch_size=B * world)
  img, _, labels = mpar.inputs(2 * B * world, S)            # type 1 feeds 2x the batch
  lams = torch.from_numpy(np.random.default_rng(4).beta(0.2, 0.2, size=B * world).astype(np.float32))
  ref = Trainer(hp, seed=0, device='cpu')
  ref.model.build((S, S))
  shard_grads = []
  for r in range(world):
    sl = slice(r * 2 * B, (r + 1) * 2 * B)
    x, onehot, _ = ref.prepare_inputs(img[sl], labels[sl], lams[r * B:(r + 1) * B])
    assert x.shape[0] == B
    ref.model(x, True, prepadded=True)
    rows, dz = ops.softmax_ce(ref.model.logits_padded, ref.model.ldc, onehot, None, B, 1001, 0.1, 0.0, 1.0, ref.model.ldc)
    ref.model.backward(dz)
    shard_grads.append(ref.model.arena.g32.clone())
  w0 = ref.model.arena.w32.clone()
  tr = Trainer(hp, seed=0, device='cpu', world_size=world)
  tr.model.build((S, S))
  tr.grad_sync = dp.GradSync(tr.model.arena, bucket_bytes=16 << 20, comm_dtype=comm_dtype)
  sl = slice(rank * 2 * B, (rank + 1) * 2 * B)
  tr.train_step(img[sl], labels[sl], lams[rank * B:(rank + 1) * B])
  g_sum = sum(shard_grads)
  if comm_dtype == 'fp32':    # four addends: the ring's summation order is not Python's, so equal up to fp32 rounding
    assert float((tr.model.arena.g32 - g_sum).norm() / g_sum.norm()) <= 1e-6
  else:   # every rank's contribution was rounded to bf16 once, the ring sums in bf16
    err = (tr.model.arena.g32 - g_sum).norm() / g_sum.norm()
    assert 0 < float(err) <= 2e-2, float(err)
  gathered = [torch.empty_like(tr.model.arena.w32) for _ in range(world)]
  dist.all_gather(gathered, tr.model.arena.w32)
  assert all(torch.equal(gathered[0], g) for g in gathered[1:]), 'replicas diverged'
  assert not torch.equal(gathered[0], w0)
  if rank == 0:
    open(os.path.join(out_dir, 'ok_' + comm_dtype), 'w').write('ok')
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('comm_dtype', ['fp32', 'bf16'])
def test_dp_world4_mixup_slices_gloo(tmp_path, comm_dtype):
  port = _free_port()
  mp.spawn(_worker_mixup, args=(4, port, str(tmp_path), comm_dtype), nprocs=4, join=True)
  assert (tmp_path / ('ok_' + comm_dtype)).exists()
