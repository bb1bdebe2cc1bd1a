"""The RCCL ('nccl' backend) gradient exchange on a real MI355X, world size 1 (gpurun exposes one GPU).

What a single rank can prove about dp.GradSync on the real library: the process group comes up over RCCL, every
bucket is handed to ncclAllReduce in watermark order while the backward tape is still running, the compute stream
waits for RCCL's stream before the optimiser reads the gradients, and the result is bit-identical to a step without
any exchange (SUM over one replica, 1/1 in the optimiser).  The 2/4/8-GPU curve is the driver's to measure."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.timeout(600)
def test_gradsync_over_rccl_world1_is_identity_and_overlapped(hip_lib):
  import torch.distributed as dist
  from assembled_cnn_amd import dp
  from assembled_cnn_amd.train import HParams, Trainer
  from tests import model_parity as mpar
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  assert not dist.is_initialized()
  dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
  try:
    hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                 use_resnet_d=True, zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01,
                 weight_decay=1e-4, batch_size=8)
    img, _, labels = mpar.inputs(8, 64)
    img, labels = img.cuda(), labels.cuda()

    plain = Trainer(hp, seed=0, device='cuda')
    with pytest.raises(RuntimeError):
      dp.GradSync(plain.model.arena)                 # not built yet: must refuse instead of cutting zero buckets
    plain.model.build((64, 64), use_resnet_d=True)
    for _ in range(2):
      plain.train_step(img, labels)

    for bucket_mb, overlap in ((4, True), (64, True), (4, False)):
      tr = Trainer(hp, seed=0, device='cuda', world_size=1)
      tr.model.build((64, 64), use_resnet_d=True)
      sync = dp.GradSync(tr.model.arena, bucket_bytes=bucket_mb << 20, overlap=overlap)
      launched = []
      orig = sync._launch

      def spy(s, i, orig=orig, tr=tr):
        launched.append((s, i, len(tr.model._ctx.tape) if (tr.model._ctx and tr.model._ctx.tape is not None) else -1))
        return orig(s, i)
      sync._launch = spy
      tr.grad_sync = sync
      for _ in range(2):
        tr.train_step(img, labels)
      torch.cuda.synchronize()
      nb = sum(len(b) for b in sync.segments)
      assert len(launched) == 2 * nb
      assert torch.equal(tr.model.arena.w32, plain.model.arena.w32), 'RCCL exchange changed the step (bucket %d MiB)' % bucket_mb
      assert torch.equal(tr.model.arena.m32, plain.model.arena.m32)
      if overlap and bucket_mb == 4:
        assert nb >= 8
        # buckets of a segment go out highest offsets first, and the first ones while the tape still has work queued
        for s in (0, 1):
          idx = [i for (ss, i, _) in launched[:nb] if ss == s]
          assert idx == sorted(idx)
      tr.model.arena.on_grad = None
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_buckets_wait_for_every_stream_that_writes_gradients(hip_lib):
  """With the weight-gradient and BigLittle side streams ON (the default), a bucket handed to RCCL must already hold the
  gradients written on those streams.  World size 1 makes the fp32 exchange an in-place identity, which would hide a bucket
  launched too early; the bf16 exchange does not: its narrowing cast reads the bucket when it is launched and the widened
  result overwrites it, so a gradient that arrived late would come back as bf16(stale value).  Expected, bit for bit:
  bf16-rounded gradients of the same step without an exchange.  4 MiB buckets (40 launches inside the backward pass),
  several steps, a batch large enough for the side streams to lag."""
  import torch.distributed as dist
  from assembled_cnn_amd import dp
  from assembled_cnn_amd.train import HParams, Trainer
  from tests import model_parity as mpar
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  assert not dist.is_initialized()
  dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
  try:
    hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                 use_resnet_d=True, zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01,
                 weight_decay=1e-4, batch_size=32)
    img, _, labels = mpar.inputs(32, 128)
    img, labels = img.cuda(), labels.cuda()
    plain = Trainer(hp, seed=0, device='cuda')
    plain.model.build((128, 128), use_resnet_d=True)
    tr = Trainer(hp, seed=0, device='cuda', world_size=1)
    tr.model.build((128, 128), use_resnet_d=True)
    assert tr.model.arena.side_stream is not None
    sync = dp.GradSync(tr.model.arena, bucket_bytes=4 << 20, comm_dtype='bf16')
    assert tr.model.arena.side_stream is not None, 'the exchange must not switch the side streams off'
    for step in range(4):
      plain._forward_backward(img, labels, None, None)
      tr._forward_backward(img, labels, None, None)
      sync(tr.model.arena.g32)
      torch.cuda.synchronize()
      want = plain.model.arena.g32.to(torch.bfloat16).float()
      got = tr.model.arena.g32
      bad = int((want != got).sum())
      if bad:
        a = plain.model.arena
        names = [n for n, sp in a.specs.items()
                 if not torch.equal(want[sp.offset:sp.offset + sp.numel], got[sp.offset:sp.offset + sp.numel])]
        raise AssertionError('step %d: %d gradient elements differ from the bf16-rounded gradients of the plain step, in %d '
                             'variables: %s' % (step, bad, len(names), names[:12]))
      # same optimiser step on both (from the plain gradients), so that the next comparison starts from equal weights
      tr.model.arena.g32.copy_(plain.model.arena.g32)
      plain._apply(None, 1.0, 1.0)
      tr.grad_sync = None
      tr._apply(None, 1.0, 1.0)
      assert torch.equal(tr.model.arena.w32, plain.model.arena.w32)
    tr.model.arena.on_grad = None
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_recorded_step_hands_buckets_to_rccl_between_tape_segments(hip_lib):
  """Trainer.capture with the exchange attached: the bucket launches of the recorded backward pass cut the launch tape into
  segments, and the replayed step hands bucket k to RCCL after segment k.  bf16 exchange (see the test above: a bucket
  launched before its gradients arrived would come back as bf16(stale value)), 4 MiB buckets, side streams on: two eager
  steps + four replayed steps on alternating batches must leave the same weights, momentum and moving statistics, bit for
  bit, as six eager steps with the same exchange."""
  import torch.distributed as dist
  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  from tests import model_parity as mpar
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  assert not dist.is_initialized()
  dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1)
  try:
    hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                 use_resnet_d=True, zero_gamma=True, learning_rate_decay_type='cosine', base_learning_rate=0.01,
                 weight_decay=1e-4, batch_size=32)
    batches = [mpar.inputs(32, 128, seed=s) for s in (1, 2)]
    batches = [(b[0].cuda(), b[2].cuda()) for b in batches]
    runs = []
    for taped in (False, True):
      tr = Trainer(hp, seed=0, device='cuda', world_size=1)
      tr.model.build((128, 128), use_resnet_d=True)
      tr.grad_sync = dp.GradSync(tr.model.arena, bucket_bytes=4 << 20, comm_dtype='bf16')
      for s in range(6):
        if taped and s == 2:
          with pytest.raises(NotImplementedError):
            tr.capture(*batches[0], warmup=0, replay='graph')
          tr.capture(*batches[0], warmup=0)
          info = ops.tape_info(tr._tape)
          assert info['segments'] == len(tr.grad_sync.recorded) + 1 and len(tr.grad_sync.recorded) >= 8, info
        tr.train_step(*batches[s % 2])
      torch.cuda.synchronize()
      a = tr.model.arena
      runs.append((a.w32.clone(), a.m32.clone(), a.state.clone()))
      if taped:
        tr.release_graph()
        tr.train_step(*batches[0])      # eager again, exchange still attached
        torch.cuda.synchronize()
      a.on_grad = None
    for p, q in zip(*runs):
      assert bool(torch.isfinite(p).all()) and torch.equal(p, q)
  finally:
    dist.destroy_process_group()
