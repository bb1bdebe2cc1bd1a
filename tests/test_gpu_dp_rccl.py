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
