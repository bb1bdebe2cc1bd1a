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


def _two_rank_worker(rank, world, port, out_dir):
  """one rank of test_two_ranks_replay_the_recorded_step_with_the_same_exchange (both ranks share the one GPU)"""
  import torch.distributed as dist
  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  from tests import model_parity as mpar
  dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
  try:
    hp = HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                 use_resnet_d=True, zero_gamma=True, learning_rate_decay_type='cosine', base_learning_rate=0.01,
                 weight_decay=1e-4, batch_size=8 * world)
    batches = [mpar.inputs(8, 64, seed=10 * rank + s) for s in (1, 2)]          # every rank its own shard
    batches = [(b[0].cuda(), b[2].cuda()) for b in batches]
    res = {}
    for taped in (False, True):
      tr = Trainer(hp, seed=0, device='cuda', world_size=world)
      tr.model.build((64, 64), use_resnet_d=True)
      tr.grad_sync = dp.GradSync(tr.model.arena, bucket_bytes=4 << 20)
      for s in range(5):
        if taped and s == 2:
          tr.capture(*batches[0], warmup=0)
          assert ops.tape_info(tr._tape)['segments'] == len(tr.grad_sync.recorded) + 1 >= 4
        tr.train_step(*batches[s % 2])
      torch.cuda.synchronize()
      a = tr.model.arena
      res['taped' if taped else 'eager'] = (a.w32.cpu().clone(), a.m32.cpu().clone(), a.state.cpu().clone())
      if taped:
        tr.release_graph()
      a.on_grad = None
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_replay_the_recorded_step_with_the_same_exchange(hip_lib, tmp_path):
  """N > 1 for real, as far as one GPU allows: two processes on the same MI355X, each with its own shard, a gloo group
  between them (RCCL refuses two ranks on one device; the exchange logic above the backend is the same).  Each rank runs
  five steps eagerly and again as two eager + three replayed steps of a recording cut at the bucket launches: weights,
  momentum and moving statistics after the replayed run equal the eager run's bit for bit on each rank, the weights of the
  two ranks equal each other (the gradients were summed over the ranks), and they differ from a single-rank run (the
  exchange did something)."""
  import torch.multiprocessing as mp
  port = _free_port()
  mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
  r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
  for r in (r0, r1):
    for p, q in zip(r['eager'], r['taped']):
      assert bool(torch.isfinite(p).all()) and torch.equal(p, q), 'replayed run differs from the eager run'
  assert torch.equal(r0['eager'][0], r1['eager'][0]) and torch.equal(r0['eager'][1], r1['eager'][1]), 'ranks drifted apart'
  assert not torch.equal(r0['eager'][2], r1['eager'][2]), 'moving statistics are per replica: the shards differ'


@pytest.mark.timeout(1200)
def test_bench_n2_rehearsal_on_one_gpu(hip_lib):
  """bench.py's REAL N > 1 code path on hardware: two ranks launched the way the driver launches
  them, sharing the one GPU this box has (--rehearsal-one-gpu: cuda:0 for both, gloo between them) -- shards, the bucket
  plan, the recorded step cut at the bucket launches, barriers, the MAX-reduce of the elapsed time, one JSON line from
  rank 0.  The value it prints is not a scaling number and says so."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), OMP_NUM_THREADS='4')
    procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '4',
                                   '--batch', '32', '--rehearsal-one-gpu', '--no-roofline', '--no-cpu-baseline'], env=env, cwd=root,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=1100) for p in procs]
  for p, (so, se) in zip(procs, outs):
    assert p.returncode == 0, se[-3000:]
  lines0 = [l for l in outs[0][0].splitlines() if l.startswith('{')]
  lines1 = [l for l in outs[1][0].splitlines() if l.startswith('{')]
  assert len(lines0) == 1 and not lines1, 'rank 0 prints exactly one JSON line, the other ranks none'
  r = json.loads(lines0[0])
  assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['config']['global_batch'] == 64 and r['config']['parallelism'] == 'dp2'
  assert r['value'] > 0 and 'REHEARSAL' in r['data']
  dp = r['dp']
  assert dp['world'] == 2 and dp['comm_dtype'] == 'fp32' and dp['buckets'] >= 5 and dp['bucket_bytes_max'] <= 32 << 20
  assert 'segments' in r['step_mode'] and 'launch tape' in r['step_mode'], r['step_mode']


def test_allreduce_bucket_on_a_callers_own_rccl_communicator(hip_lib):
  """asm_allreduce_bucket with an ncclComm_t made through librccl's C API (one rank: the sum over the replicas is the
  identity) -- what a C caller without PyTorch does.  The exchange is issued on the communicator's stream BEHIND the
  producer stream's kernels: the bucket is written by a library kernel on the producer stream right before the call, and
  what comes back must be that kernel's output (fp32 and bf16 buckets), not the bytes that were there before."""
  import ctypes as C
  from assembled_cnn_amd import lib, ops
  try:
    rccl = C.CDLL('librccl.so.1')
  except OSError:
    rccl = C.CDLL('/opt/rocm/lib/librccl.so')

  class UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]
  uid = UniqueId()
  assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
  comm = C.c_void_p()
  rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
  assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
  try:
    L = ops.L()
    prod, cs = torch.cuda.Stream(), torch.cuda.Stream()
    n = 1 << 20
    src = torch.randn(n, device='cuda')
    bucket = torch.full((n,), float('nan'), device='cuda')
    b16 = torch.full((n,), float('nan'), device='cuda', dtype=torch.bfloat16)
    torch.cuda.synchronize()
    with torch.cuda.stream(prod):
      ops.memcpy(bucket, src)                       # the "backward kernel" that writes the bucket, on the producer stream
      ops.cast_f32_to_bf16(src, b16)
    for buf, dt in ((bucket, lib.ASM_F32), (b16, lib.ASM_BF16)):
      rc = L.asm_allreduce_bucket(buf.data_ptr(), n, dt, comm, cs.cuda_stream, prod.cuda_stream)
      assert rc == 0, L.asm_last_error()
    ops.stream_join(torch.cuda.current_stream(), cs)
    torch.cuda.synchronize()
    assert torch.equal(bucket, src) and torch.equal(b16, src.to(torch.bfloat16))
    assert L.asm_allreduce_bucket(bucket.data_ptr(), n, lib.ASM_F16, comm, cs.cuda_stream, prod.cuda_stream) == lib.ASM_ENOTSUP
  finally:
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
