#!/usr/bin/env python
"""bench.py -- images/sec of the Assemble-ResNet-50 training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N=1 by default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            (N>1: one rank per GPU over RCCL)

One "step" = one pass of the whole hot path over one synthetic minibatch already resident in HBM:
mean-subtract(+mixup) -> forward -> softmax-CE(+label smoothing) -> backward -> [gradient all-reduce]
-> momentum-SGD.  Weak scaling: the per-GPU batch is fixed as N grows.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant convolution kernel class (picked from HIP-event timings of every conv
                launch in a warm-up step), timed with HIP events on the launch stream during the timed
                region; achieved = algorithmic FLOPs of that launch / its mean duration, against the
                gfx950 dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (a restatement of the reference's TF graph; TF 1.14 itself cannot run
                here) timed on this host's cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense; MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

WORKLOADS = {
    # BASELINE.json configs[1]
    'r50': dict(desc='ResNet-50 v1.5 (resnet_version=1) bf16 train', hp=dict(resnet_version=1), model='ResNet-50'),
    # BASELINE.json configs[2]  (the configuration the metric is quoted on)
    'assemble-r50': dict(desc='Assemble-ResNet-50 (BigLittle + SK + anti_alias sconv k=3 + resnet_d) bf16 train',
                         hp=dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                                 use_resnet_d=True)),
    # BASELINE.json configs[3], per-GPU shard
    'assemble-r50-mixup': dict(desc='Assemble-ResNet-50 + mixup(type 1) + label smoothing 0.1 bf16 train',
                               hp=dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                                       anti_alias_filter_size=3, use_resnet_d=True, mixup_type=1, label_smoothing=0.1)),
    # published recipe variant (scripts/train_assemble_from_scratch.sh: use_resnet_d=False)
    'assemble-r50-nod': dict(desc='Assemble-ResNet-50 (BigLittle + SK + sconv k=3, no resnet_d) bf16 train',
                             hp=dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                                     anti_alias_filter_size=3)),
    # BASELINE.json configs[4], per-GPU shard: Assemble-ResNet-152 (alpha 1, beta 2) + knowledge distillation, batch 128
    'assemble-r152-kd': dict(desc='Assemble-ResNet-152 (BigLittle alpha=1 beta=2 + SK + sconv k=3) + KD (T=1) bf16 train',
                             hp=dict(resnet_size=152, resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
                                     anti_alias_filter_size=3, bl_alpha=1, bl_beta=2, kd_temp=1.0), batch=128,
                             model='Assemble-ResNet-152'),
}


def conv_flops(key):
  """algorithmic FLOPs of one conv launch: 2 * N*Ho*Wo * K * C*R*S."""
  kind, N, H, W, Cn, K, R, S, stride = key
  Ho = H if stride == 1 else (H - 1) // stride + 1
  Wo = W if stride == 1 else (W - 1) // stride + 1
  return 2.0 * N * Ho * Wo * K * Cn * R * S


def _cpu_baseline_worker(workload, budget_s):
  """Oracle train step (fp32) on a bounded sample of the same workload -> images/sec (runs in a child)."""
  import torch
  from oracle import assembled_oracle as O
  hp = dict(WORKLOADS[workload]['hp'])
  d = hp.pop('use_resnet_d', False)
  mix = hp.pop('mixup_type', 0)
  ls = hp.pop('label_smoothing', 0.0)
  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  threads = max(1, min(avail, 32))        # more threads than that only adds OpenMP spin on a shared host
  torch.set_num_threads(threads)
  B = 4
  size = hp.pop('resnet_size', 50)
  kd = hp.pop('kd_temp', 0.0)
  m = O.Model(size, num_classes=1001, zero_gamma=True, **hp)
  st = O.TrainState(m)
  g = torch.Generator().manual_seed(0)
  x = torch.randint(0, 256, (B * (2 if mix == 1 else 1), 224, 224, 3), generator=g).float()
  x = O.mean_image_subtraction(x)
  y = torch.randint(1, 1001, (x.shape[0],), generator=g)
  lam = torch.rand(x.shape[0] // 2, generator=g) if mix else None
  if kd > 0:
    y = torch.cat([torch.nn.functional.one_hot(y, 1001).float(), torch.randn(x.shape[0], 1001, generator=g) * 3.0], 1)
  kw = dict(lr=0.1, momentum=0.9, weight_decay=1e-4, label_smoothing=ls, mixup_type=mix, lam1=lam, use_resnet_d=d,
            kd_temp=kd)
  O.train_step(st, x, y, **kw)           # warm-up (variable creation, thread pools)
  n, t0 = 0, time.time()
  while True:
    O.train_step(st, x, y, **kw)
    n += 1
    el = time.time() - t0
    if n >= 3 or el > budget_s:
      break
  return {'value': round(B * n / el, 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
          'sample': '%d training steps of batch %d at 224x224, fp32 PyTorch-CPU restatement of the TF graph '
                    '(TF 1.14 unavailable); host reports %d cpus, %d usable' % (n, B, os.cpu_count() or 0, avail)}


def cpu_baseline(workload, budget_s=20.0, hard_timeout_s=150.0):
  """Run the CPU leg in a child process with a hard wall-clock bound so it can never stall the bench."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', workload, str(budget_s)]
  env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
  try:
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout_s, env=env, cwd=ROOT)
    for line in reversed(r.stdout.strip().splitlines()):
      if line.startswith('{'):
        return json.loads(line)
    raise RuntimeError('no result: ' + r.stderr[-300:])
  except subprocess.TimeoutExpired:
    return {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': 'CPU oracle did not finish a batch-4 step sample within %.0f s on this host' % hard_timeout_s}


def main():
  if len(sys.argv) >= 3 and sys.argv[1] == '--cpu-baseline-only':
    print(json.dumps(_cpu_baseline_worker(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 20.0)), flush=True)
    return
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (BASELINE configs: 256)')
  ap.add_argument('--workload', default='assemble-r50', choices=sorted(WORKLOADS))
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-roofline', action='store_true')
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  import __graft_entry__
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus != world:
    if world == 1 and args.gpus > 1:
      raise SystemExit('--gpus %d needs a torch.distributed.run launch with %d ranks' % (args.gpus, args.gpus))
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
  torch.cuda.set_device(local_rank)
  if rank == 0:
    __graft_entry__.build()
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', rank=rank, world_size=world)
    dist.barrier()

  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  wl = WORKLOADS[args.workload]
  B = args.batch
  if 'batch' in wl and args.batch == 256:
    B = wl['batch']               # the per-GPU shard BASELINE quotes for this configuration
  hp = HParams(**dict(dict(resnet_size=50, zero_gamma=True, weight_decay=1e-4, momentum=0.9,
                           base_learning_rate=0.1 * B * world / 256, learning_rate_decay_type='fixed',
                           batch_size=B * world, dtype='bf16'), **wl['hp']))
  dev = torch.device('cuda', local_rank)
  tr = Trainer(hp, seed=0, device=dev, world_size=world)
  tr.model.build((224, 224), use_resnet_d=hp.use_resnet_d)
  if world > 1:
    tr.grad_sync = dp.GradSync(tr.model.arena)
  g = torch.Generator(device=dev).manual_seed(1 + rank)
  nin = B * 2 if hp.mixup_type == 1 else B
  images = torch.randint(0, 256, (nin, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
  labels = torch.randint(1, 1001, (nin,), generator=g, device=dev, dtype=torch.int32)
  if hp.kd_temp > 0:   # labels = concat(one-hot, teacher logits) (nets/run_loop_classification.py:90-96)
    onehot = torch.nn.functional.one_hot(labels.long(), 1001).float()
    teacher = torch.randn((nin, 1001), generator=g, device=dev) * 3.0
    labels = torch.cat([onehot, teacher], 1).contiguous()
  lam1 = tr.sample_mixup_lambdas(nin // 2) if hp.mixup_type else None

  def step():
    return tr.train_step(images, labels, lam1)

  def sync():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  dominant = None
  for i in range(args.warmup):
    last = i == args.warmup - 1
    if last and not args.no_roofline:
      timer = ops.ConvTimer()
      ops.set_conv_timer(timer)
    step()
    if last and not args.no_roofline:
      torch.cuda.synchronize()
      ops.set_conv_timer(None)
      summ = timer.summary()
      dominant = max(summ, key=lambda k: summ[k][1]) if summ else None
  timer = None
  if dominant is not None:
    timer = ops.ConvTimer(only=dominant)
    ops.set_conv_timer(timer)

  sync()
  t0 = time.time()
  for _ in range(args.steps):
    rows = step()
  sync()
  el = time.time() - t0
  ops.set_conv_timer(None)
  if world > 1:
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t)
  loss = float(tr.cross_entropy())
  if not (loss == loss) or loss > 50:
    raise SystemExit('training diverged (loss=%r): the number would be invalid' % loss)

  if rank == 0:
    out = {
        'metric': 'images/sec %s 224^2 bf16 train' % wl.get('model', 'Assemble-ResNet-50'),
        'value': round(B * world * args.steps / el, 2),
        'unit': 'images/sec',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * el / args.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic (uint8 images resident in HBM, random-init weights)',
        'config': {'workload': wl['desc'], 'per_gpu_batch': B, 'global_batch': B * world, 'image': '224x224x3',
                   'num_classes': 1001, 'parallelism': 'dp%d' % world, 'final_cross_entropy': round(loss, 4)},
    }
    if timer is not None:
      n, ms = timer.summary()[dominant]
      fl = conv_flops(dominant)
      ach = fl / (ms / n * 1e-3) / 1e12
      kname = 'conv %s N%d %dx%dx%d -> %d, %dx%d/%d' % dominant
      traffic = None
      try:  # PMC traffic of this kernel class from the committed rocprofv3 --pmc passes (profiles/)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'round1_pmc_traffic.json')))
        traffic = pmc.get(kname, {}).get('traffic_bytes')
      except (OSError, ValueError):
        pass
      out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': traffic,
                         'kernel': kname,
                         'launches_timed': n, 'avg_launch_ms': round(ms / n, 4), 'flops_per_launch': fl}
    if world == 1 and not args.no_cpu_baseline:
      try:
        out['cpu_baseline'] = cpu_baseline(args.workload)
      except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
        out['cpu_baseline'] = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': 'failed: %r' % (e,)}
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
