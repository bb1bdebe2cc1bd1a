#!/usr/bin/env python
"""bench.py -- images/sec of the Assemble-ResNet-50 training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N=1 by default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            (N>1: one rank per GPU over RCCL)

One "step" = one pass of the whole hot path over one synthetic minibatch already resident in HBM:
mean-subtract(+mixup) -> forward -> softmax-CE(+label smoothing) -> backward -> [gradient all-reduce]
-> momentum-SGD.  Weak scaling: the per-GPU batch is fixed as N grows.  Rank 0 prints ONE JSON line.

The timed region is the product as it ships: the weight gradients run beside the input-gradient chain and the big branch of
each BigLittle stage beside the little one, on side streams; and the step is RECORDED once after the warm-up
(Trainer.capture -- what a default Trainer does on its own after three eager steps) and replayed from the library's launch tape -- the same ~910 kernel launches on the same streams, issued
by one C call in 3 - 4 ms of host time instead of ~14.5 ms of Python per step, bit-identical results (tests/test_gpu_model.py,
tests/test_gpu_dp_rccl.py) -- with the loop on the trainer's stream.  `--eager` times the eager step instead (`step_mode`
says which ran; if recording fails the bench falls back to eager and says so).  Kernels of different streams share the CUs,
so the per-class figures below come from instrumented eager single-stream steps run after the timed region; `single_stream`
is the rate of the same steps with every kernel on one stream (--single-stream times that instead).  `step_detail` gives, per
timed step, the GPU time between step ends and the host time spent enqueueing the step.

Extra objects on the line:
  roofline      what BASELINE.json's north_star names: the 3x3-convolution CLASS (every 3x3 fprop, input-gradient and
                weight-gradient launch of a step, time-weighted) against the gfx950 dense bf16 MFMA peak (2.5 PFLOP/s,
                MI355X_MICROARCH.md): achieved = the class's algorithmic FLOPs / its HIP-event time.  `dominant_layer` keeps
                the single heaviest conv launch (picked in a warm-up step, timed with HIP events on the launch stream over
                the K single-stream steps), `hbm` the batch-norm family against the HBM peak; `traffic` = HBM bytes per step of the
                class from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --single-stream` (profiles/, tools/profile_round.sh).
  dp            N = 1: the step with the gradient exchange attached to an RCCL group of ONE (real bucket launches, stream
                waits, casts; no link traffic) and `exchange_ms_exposed` = that step - the plain step.  N > 1: the bucket
                plan of the run.  No N > 1 number exists in this repository until the driver's SCALE run.
  step          the whole step against its per-layer bound sum_l max(flops_l / 2.5 PFLOP/s, bytes_l / 8 TB/s)
                (every conv as fprop + dgrad + wgrad with un-fused algorithmic bytes, every batch norm
                as 3 + 5 tensor passes), plus two class figures from HIP events of three instrumented single-stream steps run
                after the timed region: the 3x3-convolution class against the MFMA peak, the 1x1-convolution class against
                its per-layer max(MFMA, HBM) bound (`conv1x1_class`, `roofline.conv1x1`), the batch-norm family against the HBM peak.
  recipe        N = 1: the published recipe (scripts/train_assemble_from_scratch.sh: mixup type 1 + label smoothing + KD +
                DropBlock with its keep_prob schedule, no resnet_d) at the same shard size on a DEFAULT Trainer, i.e. recorded
                by the trainer itself after three eager steps (DropBlock draws and gamma in static buffers); not the headline.
  cpu_baseline  the CPU oracle (a restatement of the reference's TF graph; TF 1.14 itself cannot run
                here) timed on this host's cores, BASELINE.md section 2: value = training step of the same network
                at batch 32 (median of 3), c1 = config-1 ResNet-50 64-image eval forward (median of 5).
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
PMC_FILE = 'round6_f_pmc_traffic.json'        # dominant layer, tools/conv_bench.py in isolation (round 6, tools/profile_round.sh dominant)
CLASS_TRAFFIC_FILE = 'round6_f_step_class_traffic.json'   # per-class HBM bytes per step from --pmc passes of bench.py itself
HBM_PEAK_GBS = 8000.0            # spec; MI355X_MICROARCH.md "HBM3E peak BW" (6.29 TB/s measured with a float4 copy)

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
    # the published recipe (scripts/train_assemble_from_scratch.sh:9-34): mixup type 1, label smoothing 0.1, KD at T = 1,
    # DropBlock with its keep_prob schedule, no resnet_d, cosine learning rate with 5 warm-up epochs; per-GPU shard of
    # BASELINE config 4's size (256 mixed images from 512 inputs)
    'assemble-r50-recipe': dict(desc='Assemble-ResNet-50 published recipe: BigLittle + SK + sconv k=3 + DropBlock + mixup(type 1) '
                                     '+ label smoothing 0.1 + KD (T=1) bf16 train',
                                hp=dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                                        mixup_type=1, label_smoothing=0.1, kd_temp=1.0, use_dropblock=True,
                                        learning_rate_decay_type='cosine', lr_warmup_epochs=5, train_epochs=600)),
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


def step_bound(workload, batch):
  """Per-layer roofline bound of one training step: each convolution as fprop + dgrad + wgrad (no dgrad
  for the stem) with flops = 2 N Ho Wo Co Ci k^2 and bytes = 2 (N H W Ci + N Ho Wo Co + k^2 Ci Co) each; each batch
  norm as 3 (forward) + 5 (backward) bf16 passes over its tensor; pooling / SK / loss / optimiser passes are left out
  (so the bound is a lower bound of the bound).  Shapes come from the product model's shape-only walk."""
  from assembled_cnn_amd import nn
  from assembled_cnn_amd.train import HParams
  hp = HParams(**dict(dict(resnet_size=50, zero_gamma=True), **WORKLOADS[workload]['hp']))
  m = hp.make_model(device='cpu')
  convs, bns = [], []
  orig_desc, orig_cb = nn.ConvKernel.desc, nn.conv_bn

  def desc(self, N, H, W, stride, out_f32=False, ldy=0):
    d = orig_desc(self, N, H, W, stride, out_f32, ldy)
    convs.append((d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.Ho, d.Wo, bool(self.stem)))
    return d

  def cb(ctx, x, conv, bn, stride, relu, residual=None, res_mode=0, tap_pre=None):
    out = orig_cb(ctx, x, conv, bn, stride, relu, residual, res_mode, tap_pre)
    bns.append(out.shape[0] * out.shape[1] * out.shape[2] * out.shape[3])
    return out
  import assembled_cnn_amd.model as pmodel
  nn.ConvKernel.desc, nn.conv_bn, pmodel.conv_bn = desc, cb, cb
  try:
    ctx = nn.Ctx(m.arena, True, True, 0.997, 'cpu', False, m._layers)
    m._walk(ctx, nn.Var(None, (batch, 230, 230, 4), needs_grad=False), hp.use_resnet_d, False)
  finally:
    nn.ConvKernel.desc, nn.conv_bn, pmodel.conv_bn = orig_desc, orig_cb, orig_cb
  peak_f, peak_b = MFMA_BF16_PEAK_TFLOPS * 1e12, HBM_PEAK_GBS * 1e9
  t = fl = by = fl33 = 0.0
  for (N, H, W, C, K, R, S, Ho, Wo, stem) in convs:
    if stem:   # the halo-buffer view: R = k, S = 1, "C" = 4k rounded up; algorithmic C is 3, k x k
      H, W, C, S = H - 6, W - 6, 3, R
    f = 2.0 * N * Ho * Wo * K * C * R * S
    b = 2.0 * (N * H * W * C + N * Ho * Wo * K + R * S * C * K)
    passes = 2 if stem else 3
    t += passes * max(f / peak_f, b / peak_b)
    fl += passes * f
    by += passes * b
    if R == 3 and S == 3:
      fl33 += passes * f
  bn_bytes = sum(bns) * 2.0 * 8
  t += bn_bytes / peak_b
  by += bn_bytes
  return {'bound_ms': t * 1e3, 'flops': fl, 'bytes': by, 'flops_3x3': fl33, 'bn_bytes': bn_bytes}


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
  # BASELINE.md section 2 asks for all usable cores; on the 256-logical-CPU GPU host 256 OpenMP threads did not finish a
  # batch-32 step in 200 s (oversubscribed SMT siblings spinning at every barrier), so the pool is capped at 64
  # threads and the line says how many were used out of how many usable.
  threads = max(1, min(avail, int(os.environ.get('ASM_CPU_BASELINE_THREADS', '64'))))
  torch.set_num_threads(threads)
  import statistics
  B = 32
  size = hp.pop('resnet_size', 50)
  kd = hp.pop('kd_temp', 0.0)
  g = torch.Generator().manual_seed(0)
  # C1: BASELINE config 1 -- ResNet-50 resnet_version=1, 64 seeded 224 x 224 images, eval forward, median of 5
  t_c1 = time.time()
  m1 = O.Model(50, num_classes=1001)
  x1 = O.mean_image_subtraction(torch.randint(0, 256, (64, 224, 224, 3), generator=g).float())
  with torch.no_grad():
    m1(x1[:2], False)
    m1(x1, False)
    c1 = []
    for _ in range(5):
      t0 = time.time()
      m1(x1, False)
      c1.append(time.time() - t0)
      if time.time() - t_c1 > budget_s:
        break
  c1_ips = 64.0 / statistics.median(c1)
  del m1, x1
  # C2-style: one full training step (fwd + bwd + momentum update) of THIS workload's network at batch 32, fp32
  m = O.Model(size, num_classes=1001, zero_gamma=True, **hp)
  st = O.TrainState(m)
  x = torch.randint(0, 256, (B * (2 if mix == 1 else 1), 224, 224, 3), generator=g).float()
  x = O.mean_image_subtraction(x)
  y = torch.randint(1, 1001, (x.shape[0],), generator=g)
  lam = torch.rand(x.shape[0] // 2, generator=g) if mix else None
  if kd > 0:
    y = torch.cat([torch.nn.functional.one_hot(y, 1001).float(), torch.randn(x.shape[0], 1001, generator=g) * 3.0], 1)
  kw = dict(lr=0.1, momentum=0.9, weight_decay=1e-4, label_smoothing=ls, mixup_type=mix, lam1=lam, use_resnet_d=d,
            kd_temp=kd)
  O.train_step(st, x, y, **kw)           # warm-up (variable creation, thread pools)
  ts, t_all = [], time.time()
  while len(ts) < 3:
    t0 = time.time()
    O.train_step(st, x, y, **kw)
    ts.append(time.time() - t0)
    if time.time() - t_all > budget_s:
      break
  return {'value': round(B / statistics.median(ts), 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
          'c1_eval_images_per_sec': round(c1_ips, 2),
          'sample': 'median of %d training steps of batch %d at 224x224 of the same network (fp32 PyTorch-CPU restatement '
                    'of the TF graph; TF 1.14 unavailable) + c1: ResNet-50 v1.5 64-image eval forward, median of %d; '
                    'host reports %d cpus, %d usable, %d torch threads' % (len(ts), B, len(c1), os.cpu_count() or 0, avail, threads)}


def cpu_baseline(workload, budget_s=25.0, hard_timeout_s=200.0):
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


def _flush_c_stdio():
  try:
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
  except Exception:
    pass


def _dp_plan(gs, world, backend='RCCL (torch.distributed nccl)'):
  nb = sum(len(seg) for seg in gs.segments)
  width = 2 if gs.comm_dtype == 'bf16' else 4
  return {'world': world, 'buckets': nb, 'bucket_bytes_max': max((hi - lo) for seg in gs.segments for lo, hi in seg) * width,
          'bytes_per_step': gs.arena.total_elems * width, 'comm_dtype': gs.comm_dtype, 'backend': backend,
          'ring_bytes_per_gpu_per_step': int(2 * (world - 1) / max(world, 1) * gs.arena.total_elems * width),
          'overlap': 'buckets are launched as the backward watermark passes them; the optimiser waits for the last one'}


def _gradsync_leg(tr, step, sync, args, plain_ms, capture=None):
  """N = 1: the same steps with dp.GradSync attached to an RCCL group of ONE rank: every bucket launch, stream wait and
  (bf16) cast is real, only the link traffic is missing.  exchange_ms_exposed = step with the exchange - step without."""
  import socket
  import torch.distributed as dist
  from assembled_cnn_amd import dp
  try:
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    if os.environ.get('NCCL_DEBUG', '').upper() in ('', 'VERSION'):
      os.environ['NCCL_DEBUG'] = 'WARN'     # no version banner on stdout next to the one JSON line
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    gs = dp.GradSync(tr.model.arena, comm_dtype=args.comm_dtype)
    tr.grad_sync = gs
    if capture is not None:   # the timed region replayed a recorded step: so does this leg (the tape cut at the bucket launches)
      capture()
    for _ in range(2):
      step()
    sync()
    t1 = time.time()
    for _ in range(args.steps):
      step()
    sync()
    ms = 1000.0 * (time.time() - t1) / args.steps
    info = _dp_plan(gs, 1)
    info.update({'ms_per_step_with_exchange': round(ms, 3), 'exchange_ms_exposed': round(ms - plain_ms, 3),
                 'what': 'RCCL group of ONE rank on this GPU (bucket launches, waits and casts are real, xGMI traffic is '
                         'not); NO N > 1 number exists in this repository until the driver\'s SCALE run'})
    if capture is not None:
      info['step_mode'] = 'launch tape in %d segments, one bucket handed to RCCL after each but the last' % (len(gs.recorded) + 1)
      tr.release_graph()
    tr.grad_sync = None
    tr.model.arena.on_grad = None
    dist.destroy_process_group()
    return info
  except Exception as e:   # a reported extra must never lose the measured number
    tr.grad_sync = None
    tr.model.arena.on_grad = None
    try:
      tr.release_graph()
    except Exception:
      pass
    return {'world': 1, 'error': repr(e)}


def _recipe_leg(Trainer, make_hp, make_inputs, dev, args, B):
  """N = 1: the published recipe (scripts/train_assemble_from_scratch.sh: mixup type 1 + label smoothing + KD + DropBlock with
  its keep_prob schedule) on a DEFAULT Trainer -- nobody calls capture(): train_step records itself after its three eager
  steps and replays from then on, DropBlock draws and gamma rewritten in their static buffers before every replay."""
  import torch
  try:
    wl = WORKLOADS['assemble-r50-recipe']
    hp = make_hp(wl)
    tr = Trainer(hp, seed=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    images, labels, nin = make_inputs(hp, g)
    lam1 = tr.sample_mixup_lambdas(nin // 2)
    steps = min(args.steps, 30)
    with torch.cuda.stream(tr.stream):
      for _ in range(Trainer.AUTO_WARMUP + 3):
        tr.train_step(images, labels, lam1)
      torch.cuda.synchronize()
      t0 = time.time()
      for _ in range(steps):
        tr.train_step(images, labels, lam1)
      torch.cuda.synchronize()
      el = time.time() - t0
    loss = float(tr.cross_entropy())
    out = {'workload': wl['desc'], 'value': round(B * steps / el, 2), 'unit': 'images/sec', 'ms_per_step': round(1000.0 * el / steps, 3),
           'steps': steps, 'step_mode': tr.step_mode, 'keep_prob': round(float(tr.last.get('keep_prob', 1.0)), 6),
           'final_cross_entropy': round(loss, 4),
           'what': 'per-GPU shard of BASELINE config 4 (512 uint8 images -> 256 mixed) + KD + DropBlock on a default Trainer: '
                   'the step records itself after %d eager steps (no capture() call); not the headline configuration' % Trainer.AUTO_WARMUP}
    tr.release_graph()
    del tr
    torch.cuda.empty_cache()
    return out
  except Exception as e:   # a reported extra must never lose the measured number
    return {'error': repr(e)}


def main():
  if len(sys.argv) >= 3 and sys.argv[1] == '--cpu-baseline-only':
    print(json.dumps(_cpu_baseline_worker(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 20.0)), flush=True)
    return
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)       # >= 50 timed steps after >= 10 warm-up steps
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (BASELINE configs: 256)')
  ap.add_argument('--workload', default='assemble-r50', choices=sorted(WORKLOADS))
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-roofline', action='store_true')
  ap.add_argument('--single-stream', action='store_true',
                  help='time the step with every kernel on ONE stream (ASM_WGRAD_STREAM=0 ASM_BL_STREAMS=0); without it the '
                       'timed region is the product default (weight gradients and the big branch of a BigLittle '
                       'stage on side streams) and the single-stream rate is reported as an extra')
  ap.add_argument('--dump-convs', default='', help='write the per-conv-shape HIP-event times of the instrumented step here (markdown)')
  ap.add_argument('--eager', action='store_true',
                  help='N = 1: time the eager step (~930 launches enqueued by Python) instead of the recorded one '
                       '(Trainer.capture: the same launches replayed from a launch tape by one C call)')
  ap.add_argument('--no-gradsync', action='store_true', help='N = 1: skip the extra leg with the gradient exchange attached')
  ap.add_argument('--no-recipe', action='store_true',
                  help='N = 1: skip the extra leg that times the published recipe (mixup + label smoothing + KD + DropBlock) as the '
                       'step a default Trainer records on its own')
  ap.add_argument('--rehearsal-one-gpu', action='store_true',
                  help='N > 1 REHEARSAL on a box with one GPU: every rank uses cuda:0 and the process group is gloo (RCCL refuses '
                       'two ranks on one device).  Executes the real N > 1 control flow of this script -- shards, bucket plan, '
                       'recorded step cut at the bucket launches, barriers, MAX-reduce -- on hardware; the value is NOT a scaling number')
  ap.add_argument('--comm-dtype', default='fp32', choices=['fp32', 'bf16'], help='precision of the exchanged gradient buckets')
  ap.add_argument('--dry-run-cpu', action='store_true',
                  help='TEST ONLY (tests/test_bench_dryrun_cpu.py): run the control flow of this script -- rank-0 build, '
                       'rendezvous, GradSync, barriers, MAX-reduce of the elapsed time, the JSON line -- on CPU over gloo with '
                       'the test double of the C ABI and a tiny batch; the number it prints means nothing')
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
  dry = args.dry_run_cpu
  if not dry:
    if not torch.cuda.is_available():
      raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    if args.rehearsal_one_gpu:
      local_rank = 0
    torch.cuda.set_device(local_rank)
  if rank == 0:
    __graft_entry__.build()
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo' if (dry or args.rehearsal_one_gpu) else 'nccl', rank=rank, world_size=world)
    dist.barrier()
  if dry:
    args.no_roofline = args.no_cpu_baseline = args.no_gradsync = True
    from assembled_cnn_amd import ops as _ops
    from tests.cpu_double import CpuDouble      # test infrastructure, only under --dry-run-cpu
    _ops.set_library(CpuDouble(), is_double=True)

  # The timed region is the product as it ships: weight gradients beside the dgrad chain and the big branch of a BigLittle
  # stage beside the little one on side streams (dp.GradSync joins them before each bucket launch).  Kernels then share the CUs and their individual durations depend on what runs next to
  # them, so the per-class HIP-event sums (`roofline`, `step`) come from instrumented SINGLE-stream steps after the timed
  # region, where a duration is a property of the kernel; the single-stream rate is reported as `single_stream`.
  if args.single_stream:
    os.environ['ASM_WGRAD_STREAM'] = '0'
    os.environ['ASM_BL_STREAMS'] = '0'
  from assembled_cnn_amd import dp, ops
  from assembled_cnn_amd.train import HParams, Trainer
  wl = WORKLOADS[args.workload]
  B = args.batch
  if 'batch' in wl and args.batch == 256:
    B = wl['batch']               # the per-GPU shard BASELINE quotes for this configuration
  def make_hp(w):
    return HParams(**dict(dict(resnet_size=50, zero_gamma=True, weight_decay=1e-4, momentum=0.9,
                               base_learning_rate=0.1 * B * world / 256, learning_rate_decay_type='fixed',
                               batch_size=B * world, dtype='bf16'), **w['hp']))

  def make_inputs(hp_, gen):
    nin_ = B * 2 if hp_.mixup_type == 1 else B
    im = torch.randint(0, 256, (nin_, side, side, 3), generator=gen, device=dev, dtype=torch.uint8)
    lb = torch.randint(1, 1001, (nin_,), generator=gen, device=dev, dtype=torch.int32)
    if hp_.kd_temp > 0:   # labels = concat(one-hot, teacher logits) (nets/run_loop_classification.py:90-96)
      oh = torch.nn.functional.one_hot(lb.long(), 1001).float()
      te = torch.randn((nin_, 1001), generator=gen, device=dev) * 3.0
      lb = torch.cat([oh, te], 1).contiguous()
    return im, lb, nin_

  hp = make_hp(wl)
  dev = torch.device('cpu') if dry else torch.device('cuda', local_rank)
  side = 64 if dry else 224
  # recorded=False: this script decides itself when the step is recorded (after the warm-up) and releases the recording for
  # its instrumented legs; a default Trainer records on its own after three eager steps (the `recipe` leg below uses one)
  tr = Trainer(hp, seed=0, device=dev, world_size=world, recorded=False if not wl['hp'].get('use_dropblock') else None)
  if wl['hp'].get('use_dropblock'):
    tr._auto = False            # DropBlock needs the trainer's static draw buffers; the recording is still made below
  tr.model.build((side, side), use_resnet_d=hp.use_resnet_d)
  dp_info = None
  if world > 1:
    tr.grad_sync = dp.GradSync(tr.model.arena, comm_dtype=args.comm_dtype)
    dp_info = _dp_plan(tr.grad_sync, world, 'gloo (REHEARSAL on one GPU)' if args.rehearsal_one_gpu else ('gloo (dry run)' if dry else 'RCCL (torch.distributed nccl)'))
  g = torch.Generator(device=dev).manual_seed(1 + rank)
  images, labels, nin = make_inputs(hp, g)
  lam1 = tr.sample_mixup_lambdas(nin // 2) if hp.mixup_type else None

  def step():
    return tr.train_step(images, labels, lam1)

  def sync():
    if not dry:
      torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      if not dry:
        torch.cuda.synchronize()

  def pick_dominant():
    """one step with HIP events around every conv launch -> the (kind, shape) with the most time"""
    t = ops.ConvTimer()
    ops.set_conv_timer(t)
    step()
    torch.cuda.synchronize()
    ops.set_conv_timer(None)
    summ = t.summary()
    return max(summ, key=lambda k: summ[k][1]) if summ else None

  # N = 1, product default: the step as a launch tape (Trainer.capture) -- bit-identical to the eager step, ~3 ms of host
  # time per step instead of ~14, so a busy host cannot make the step host-bound.  (N > 1: the tape is cut where the host
  # hands a gradient bucket to RCCL.  --single-stream: eager, its HIP-event instrumentation wraps launches.)
  taped = not dry and not args.eager and not args.single_stream and os.environ.get('ASM_STEP_TAPE', '1') != '0'
  on_stream = (taped and os.environ.get('ASM_BENCH_STREAM', '1') != '0') or (not dry and os.environ.get('ASM_BENCH_STREAM', '') == '1')
  if on_stream:
    # The loop itself runs on the trainer's stream (`with torch.cuda.stream(trainer.stream)` in a training script): a
    # recorded step replayed from the default stream joins that stream with its own at both ends of every step, which
    # measured 25.9 ms per step against 24.65 with the default stream out of the loop (same box, same recording).
    tr.stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(tr.stream)
  dominant = None
  for i in range(args.warmup):
    if i == args.warmup - 1 and not args.no_roofline and args.single_stream:
      dominant = pick_dominant()
    else:
      step()
  # eager step only: stream calibration (untimed, after the warm-up): Trainer.calibrate_streams keeps the side streams unless
  # they are slower than one stream in this process (a busy host: they cost more host calls); ASM_STREAM_AUTOTUNE=0 skips it
  stream_cal = None
  if not dry and not args.single_stream and not taped and os.environ.get('ASM_STREAM_AUTOTUNE', '1') != '0':
    stream_cal = tr.calibrate_streams(step)
  tape_info = tape_error = None

  def capture_step():
    tr.capture(images, labels, lam1, warmup=0, replay=os.environ.get('ASM_STEP_REPLAY', 'tape'))

  if taped:
    try:
      capture_step()
      tape_info = ops.tape_info(tr._tape) if tr._tape is not None else {'launches': 0, 'joins': 0, 'segments': 1}
      step()                      # first replay (untimed)
    except Exception as e:        # the number must not depend on the recording: fall back to the eager step, and say so
      tape_error = repr(e)
      taped = False
      try:
        tr.release_graph()
      except Exception:
        pass
      torch.cuda.synchronize()
  timer = None
  if dominant is not None and args.single_stream:
    timer = ops.ConvTimer(only=dominant)
    ops.set_conv_timer(timer)

  sync()
  calls0 = ops.abi_calls()
  kern0 = ops.L().asm_launch_count() if not dry else 0
  # per-step diagnostics that cost two clock reads and one event record per step: how long the HOST took to enqueue each
  # step and when the GPU finished it -- a timed region that is slow because the host fell behind (a busy box) reads
  # differently from one where the GPU ran slowly, and either shows as its first / last steps
  step_host, step_ev = [], []
  t0 = time.time()
  for _ in range(args.steps):
    h0 = time.perf_counter()
    rows = step()
    if not dry:
      ev = torch.cuda.Event(enable_timing=True)
      ev.record()
      step_ev.append(ev)
    step_host.append(time.perf_counter() - h0)
  sync()
  el = time.time() - t0
  abi_calls = (ops.abi_calls() - calls0) / max(args.steps, 1)
  kernels_per_step = ((ops.L().asm_launch_count() - kern0) / max(args.steps, 1)) if not dry else None
  ops.set_conv_timer(None)
  if world > 1:
    # An N-rank line a reader can verify at a glance: who took part (gathered over the process group itself), what each
    # rank's own clock said, and how much of the gradient exchange was NOT hidden behind the backward pass at this N --
    # the same K steps again in the same step mode with the exchange detached (ranks then drift apart: the leg comes after
    # the timed region and nothing measured afterwards depends on the weights).
    import socket
    mine = {'rank': rank, 'local_rank': local_rank, 'host': socket.gethostname(),
            'device': 'cpu' if dry else '%s #%d (%s)' % (torch.cuda.get_device_name(dev), dev.index,
                                                         getattr(torch.cuda.get_device_properties(dev), 'pci_bus_id', '?')),
            'ms_per_step': round(1000.0 * el / args.steps, 3)}
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    plain_ms, leg_s = None, -1.0
    try:      # no collective inside this block: a rank that fails here must not leave the others waiting in one
      if taped:
        tr.release_graph()
      tr.grad_sync = None
      tr.model.arena.on_grad = None
      if taped:
        capture_step()
      step()
      if not dry:
        torch.cuda.synchronize()
      t1 = time.time()
      for _ in range(args.steps):
        step()
      if not dry:
        torch.cuda.synchronize()
      leg_s = time.time() - t1
      if taped:
        tr.release_graph()
    except Exception as e:       # a reported extra must never lose the measured number
      mine['exchange_leg_error'] = repr(e)
    tp = torch.tensor([leg_s, -leg_s], device=dev, dtype=torch.float64)     # every rank reaches this: MAX time, MIN time (< 0: a rank failed)
    dist.all_reduce(tp, op=dist.ReduceOp.MAX)
    if -float(tp[1]) > 0:
      plain_ms = 1000.0 * float(tp[0]) / args.steps
    ms_all = [r_['ms_per_step'] for r_ in seen if r_ and 'ms_per_step' in r_]
    dp_info.update({'world': world, 'world_seen_by_backend': dist.get_world_size(), 'backend_name': dist.get_backend(),
                    'ranks': seen, 'rank_ms_per_step_min': min(ms_all), 'rank_ms_per_step_max': max(ms_all),
                    'ms_per_step_without_exchange': None if plain_ms is None else round(plain_ms, 3),
                    'exchange_ms_exposed': None if plain_ms is None else round(max(ms_all) - plain_ms, 3),
                    'what': 'ranks: gathered with all_gather_object over the group the gradients were exchanged on; '
                            'exchange_ms_exposed = the timed region (max over ranks) minus the same %d steps, same step mode, '
                            'with the exchange detached' % args.steps})
  eager_leg = None
  if taped:
    tr.release_graph()          # the legs below instrument or re-wire the eager step
    if world == 1:              # the same K steps enqueued launch by launch by the Python host code, same streams: for the record
      step()
      sync()
      t1 = time.time()
      for _ in range(args.steps):
        step()
      sync()
      el_e = time.time() - t1
      eager_leg = {'value': round(B * world * args.steps / el_e, 2), 'ms_per_step': round(1000.0 * el_e / args.steps, 3),
                   'what': 'the same %d steps as EAGER steps (every launch enqueued by Python, ~14.5 ms of host time per step), '
                           'same streams, right after the timed region: the recorded step launches the same kernels, so the two '
                           'agree when the host keeps up and part ways when it does not' % args.steps}
  if world > 1:
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t)
  if world == 1 and not args.no_gradsync and not dry:   # same streams as the timed region, plus the exchange
    dp_info = _gradsync_leg(tr, step, sync, args, 1000.0 * el / args.steps, capture_step if taped else None)
  if on_stream:
    torch.cuda.default_stream().wait_stream(tr.stream)
    torch.cuda.set_stream(torch.cuda.default_stream())
  recipe = None
  if world == 1 and not dry and not args.no_recipe and args.workload != 'assemble-r50-recipe':
    recipe = _recipe_leg(Trainer, make_hp, make_inputs, dev, args, B)
  class_sum = None
  INSTR = 3
  single = None
  if not dry and not args.single_stream and not args.no_roofline:
    # everything below runs on ONE stream
    os.environ['ASM_BL_STREAMS'] = '0'
    ops.refresh_tuning()
    tr.model.arena.disable_side_stream()
    if world == 1:
      step()
      dominant = pick_dominant() if not args.no_roofline else None      # picked on one stream: durations are the kernels' own
      if dominant is not None:      # the heaviest layer's launches, HIP events on the launch stream over this leg
        timer = ops.ConvTimer(only=dominant)
        ops.set_conv_timer(timer)
      sync()
      t1 = time.time()
      for _ in range(args.steps):
        step()
      sync()
      el1 = time.time() - t1
      ops.set_conv_timer(None)
      single = {'value': round(B * world * args.steps / el1, 2), 'ms_per_step': round(1000.0 * el1 / args.steps, 3),
                'what': 'the same %d steps with every kernel on one HIP stream (ASM_WGRAD_STREAM=0 ASM_BL_STREAMS=0): the '
                        'state the per-class HIP-event sums and the rocprofv3 summaries under profiles/ describe' % args.steps}
  if not args.no_roofline:   # instrumented steps after the timed region: HIP events around every conv and BN-family call
    step()
    ct = ops.ConvTimer(classes=True)
    ops.set_conv_timer(ct)
    for _ in range(INSTR):
      step()
    torch.cuda.synchronize()
    ops.set_conv_timer(None)
    class_sum = ({k: (v[0] / INSTR, v[1] / INSTR) for k, v in ct.summary().items()},
                 {k: (v[0] / INSTR, v[1] / INSTR, v[2] / INSTR) for k, v in ct.class_summary().items()})
    if args.dump_convs and rank == 0:
      rows = sorted(((v[1], k, v[0]) for k, v in class_sum[0].items()), reverse=True)
      with open(args.dump_convs, 'w') as f:
        f.write('| kind | N HxWxC -> K, RxS/stride | launches | ms per step | TFLOP/s | GB/s (in + out once) |\n|---|---|---:|---:|---:|---:|\n')
        for ms, k, n in rows:
          kind, N_, H_, W_, C_, K_, R_, S_, st_ = k
          Ho_ = H_ if st_ == 1 else (H_ - 1) // st_ + 1
          by = 2.0 * N_ * (H_ * W_ * C_ + Ho_ * Ho_ * K_) * n
          f.write('| %s | %d %dx%dx%d -> %d, %dx%d/%d | %d | %.4f | %.0f | %.0f |\n' % (
              kind, N_, H_, W_, C_, K_, R_, S_, st_, n, ms, conv_flops(k) * n / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9))
  loss = float(tr.cross_entropy())
  if not (loss == loss) or loss > 50:
    raise SystemExit('training diverged (loss=%r): the number would be invalid' % loss)

  if world > 1:       # whatever a native library buffered on stdout (an RCCL banner) goes out BEFORE rank 0's JSON line
    _flush_c_stdio()
    dist.barrier()
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
    dominant_obj = None
    if timer is not None:
      n, ms = timer.summary()[dominant]
      fl = conv_flops(dominant)
      ach = fl / (ms / n * 1e-3) / 1e12
      kname = 'conv %s N%d %dx%dx%d -> %d, %dx%d/%d' % dominant
      traffic = None
      try:  # PMC traffic of this kernel class from the committed rocprofv3 --pmc passes (profiles/)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', PMC_FILE)))
        traffic = pmc.get(kname, {}).get('traffic_bytes')
      except (OSError, ValueError):
        pass
      dominant_obj = {'kernel': kname, 'achieved': round(ach, 2), 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                      'launches_timed': n, 'avg_launch_ms': round(ms / n, 4), 'flops_per_launch': fl, 'traffic': traffic,
                      'traffic_source': 'profiles/%s (rocprofv3 --pmc passes over tools/conv_bench.py of this layer)' % PMC_FILE,
                      'timed': 'HIP events on the launch stream around every launch of this layer over the %d steps of the '
                               '%s' % (args.steps, 'timed region (single stream)' if args.single_stream else
                                       'single-stream leg (`single_stream`): beside another stream\'s kernels a duration '
                                       'is not a property of the kernel')}
    try:
      sb = step_bound(args.workload, B)
      ms_step = 1000.0 * el / args.steps
      st_obj = {'bound_ms': round(sb['bound_ms'], 3), 'frac_of_bound': round(sb['bound_ms'] / ms_step, 4),
                'algorithmic_tflop': round(sb['flops'] / 1e12, 3), 'algorithmic_gb': round(sb['bytes'] / 1e9, 2),
                'achieved_tflops': round(sb['flops'] / (ms_step * 1e-3) / 1e12, 1),
                'peaks': {'mfma_tflops': MFMA_BF16_PEAK_TFLOPS, 'hbm_gbs': HBM_PEAK_GBS}}
      if class_sum is not None:
        convs, classes = class_sum
        f33 = sum(conv_flops(k) * v[0] for k, v in convs.items() if k[6] == 3 and k[7] == 3)
        t33 = sum(v[1] for k, v in convs.items() if k[6] == 3 and k[7] == 3)
        tall = sum(v[1] for v in convs.values())
        # algorithmic bytes of the class: every launch reads its input and filter once and writes its output once (bf16;
        # a weight gradient reads x and dy and writes fp32 dW) -- what `traffic` (counter bytes) is to be held against
        def conv_bytes(k):
          kind, N_, H_, W_, C_, K_, R_, S_, st_ = k
          Ho_ = H_ if st_ == 1 else (H_ - 1) // st_ + 1
          Wo_ = W_ if st_ == 1 else (W_ - 1) // st_ + 1
          io = 2.0 * (N_ * H_ * W_ * C_ + N_ * Ho_ * Wo_ * K_)
          return io + (4.0 if kind == 'wgrad' else 2.0) * R_ * S_ * C_ * K_
        b33 = sum(conv_bytes(k) * v[0] for k, v in convs.items() if k[6] == 3 and k[7] == 3)
        st_obj['conv3x3_class'] = {'ms_per_step': round(t33, 3), 'tflops': round(f33 / (t33 * 1e-3) / 1e12, 1),
                                   'algorithmic_bytes_per_step': int(b33),
                                   'frac_of_mfma_peak': round(f33 / (t33 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                   'launches': sum(v[0] for k, v in convs.items() if k[6] == 3 and k[7] == 3)}
        f11 = sum(conv_flops(k) * v[0] for k, v in convs.items() if k[6] == 1 and k[7] == 1 and k[2] > 1)
        t11 = sum(v[1] for k, v in convs.items() if k[6] == 1 and k[7] == 1 and k[2] > 1)
        b11 = sum(conv_bytes(k) * v[0] for k, v in convs.items() if k[6] == 1 and k[7] == 1 and k[2] > 1)
        if t11 > 0:
          # the 1x1 class (conv1 / conv3 / projections on maps larger than 1 x 1) is bandwidth-bound on the large maps and
          # matrix-bound on the small ones: both fractions are given; its bound is the per-layer max of the two
          st_obj['conv1x1_class'] = {'ms_per_step': round(t11, 3), 'tflops': round(f11 / (t11 * 1e-3) / 1e12, 1),
                                     'algorithmic_bytes_per_step': int(b11), 'gbs_in_out_once': round(b11 / (t11 * 1e-3) / 1e9, 1),
                                     'frac_of_mfma_peak': round(f11 / (t11 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                     'frac_of_hbm_peak': round(b11 / (t11 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     'bound_ms': round(sum(max(conv_flops(k) / (MFMA_BF16_PEAK_TFLOPS * 1e12), conv_bytes(k) / (HBM_PEAK_GBS * 1e9))
                                                           * v[0] for k, v in convs.items() if k[6] == 1 and k[7] == 1 and k[2] > 1) * 1e3, 3),
                                     'launches': sum(v[0] for k, v in convs.items() if k[6] == 1 and k[7] == 1 and k[2] > 1),
                                     'note': 'in + out once: the fan-in addend a conv1 input gradient also reads is not counted'}
        st_obj['conv_all_ms_per_step'] = round(tall, 3)
        if 'bn' in classes:
          nb, tb, wb = classes['bn']
          st_obj['bn_class'] = {'ms_per_step': round(tb, 3), 'algorithmic_gb': round(wb / 1e9, 2),
                                'gbs': round(wb / (tb * 1e-3) / 1e9, 1),
                                'frac_of_hbm_peak': round(wb / (tb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'calls': nb}
        st_obj['class_source'] = 'HIP events around every conv / batch-norm-family call of %d instrumented steps after the timed region' % INSTR
        cls_traffic = {}
        try:
          cls_traffic = json.load(open(os.path.join(ROOT, 'profiles', CLASS_TRAFFIC_FILE)))
        except (OSError, ValueError):
          pass
        c33 = st_obj['conv3x3_class']
        out['roofline'] = {
            'bound': 'mfma', 'achieved': c33['tflops'], 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': c33['frac_of_mfma_peak'], 'traffic': cls_traffic.get('conv3x3_class_bytes_per_step'),
            'algorithmic_bytes': c33['algorithmic_bytes_per_step'],
            'kernel': '3x3 convolution class: every 3x3 fprop / input-gradient / weight-gradient launch of a step (%d launches, '
                      '%.3f ms, %.1f algorithmic GFLOP), time-weighted' % (c33['launches'], c33['ms_per_step'], f33 / 1e9),
            'traffic_source': 'profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --single-stream`, '
                              'tools/profile_round.sh; not measured in this run)' % CLASS_TRAFFIC_FILE,
            'dominant_layer': dominant_obj}
        if 'conv1x1_class' in st_obj:
          c11 = st_obj['conv1x1_class']
          out['roofline']['conv1x1'] = {'bound': 'hbm / mfma per layer', 'ms_per_step': c11['ms_per_step'], 'bound_ms': c11['bound_ms'],
                                        'frac': round(c11['bound_ms'] / c11['ms_per_step'], 4), 'achieved_tflops': c11['tflops'],
                                        'achieved_gbs': c11['gbs_in_out_once'],
                                        'kernel': '1x1 convolution class: every 1x1 fprop / input-gradient / weight-gradient launch on maps '
                                                  'larger than 1 x 1 (%d launches); frac = sum of per-layer max(flops / 2.5 PFLOP/s, '
                                                  'bytes / 8 TB/s) over the class time' % c11['launches']}
        if 'bn_class' in st_obj:
          bc = st_obj['bn_class']
          out['roofline']['hbm'] = {'bound': 'hbm', 'achieved': bc['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'frac': bc['frac_of_hbm_peak'], 'traffic': cls_traffic.get('bn_class_bytes_per_step'),
                                    'kernel': 'batch-norm family (statistics finalize, apply, backward reduce / apply, the SK '
                                              'unit\'s on-the-fly forms): %.2f algorithmic GB in %.3f ms per step' % (
                                                  bc['algorithmic_gb'], bc['ms_per_step'])}
      elif dominant_obj is not None:
        out['roofline'] = dict({'bound': 'mfma', 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s'}, **dominant_obj)
      out['step'] = st_obj
    except Exception as e:   # reporting extras must never lose the measured number
      out['step'] = {'error': repr(e)}
    out['streams'] = ('single (ASM_WGRAD_STREAM=0 ASM_BL_STREAMS=0)' if args.single_stream else
                      'product default: the weight gradients and the big branch of each BigLittle stage on side streams')
    if len(step_ev) >= 2:
      gaps = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(len(step_ev) - 1))
      hs = sorted(1000.0 * h for h in step_host)
      out['step_detail'] = {'gpu_ms_between_step_ends': {'min': round(gaps[0], 3), 'median': round(gaps[len(gaps) // 2], 3),
                                                         'max': round(gaps[-1], 3)},
                            'host_enqueue_ms': {'min': round(hs[0], 3), 'median': round(hs[len(hs) // 2], 3),
                                                'max': round(hs[-1], 3)},
                            'host': {'cpus': os.cpu_count(), 'loadavg_1min': round(os.getloadavg()[0], 2)},
                            'what': 'per timed step: time between consecutive end-of-step events on the compute stream, and '
                                    'host time to enqueue the step (the host runs ahead of the GPU when its median is the smaller)'}
    out['step_mode'] = ('launch tape: the step recorded once by Trainer.capture and replayed by asm_tape_replay (%d kernel launches, '
                        '%d cross-stream joins per step; bit-identical to the eager step, tests/test_gpu_model.py)'
                        % (tape_info['launches'], tape_info['joins'])) if tape_info else 'eager: every launch enqueued by the Python host code'
    if tape_info and tape_info.get('segments', 1) > 1:
      out['step_mode'] += '; %d segments, a gradient bucket handed to RCCL after each but the last' % tape_info['segments']
    if tape_error:
      out['step_mode'] += ' (recording the step failed: %s)' % tape_error
    if eager_leg is not None:
      out['eager_step'] = eager_leg
    if stream_cal is not None:
      out['streams_autotune'] = dict(stream_cal, what='Trainer.calibrate_streams after the warm-up, 3 untimed steps per setting: '
                                     'the timed region runs the chosen one (side streams unless > 3 % slower than one stream, which happens when the host is busy)')
      if stream_cal['chosen'] != 'side streams':
        out['streams'] = 'single stream (chosen by Trainer.calibrate_streams: the side streams measured slower in this process)' 
    out['launches'] = {'kernels_per_step': None if kernels_per_step is None else round(kernels_per_step, 1),
                       'abi_calls_per_step': round(abi_calls, 1),
                       'note': 'kernels_per_step: the library\'s own launch counter (asm_launch_count) across the timed region, '
                               'every stream; torch launches nothing inside a step.  abi_calls_per_step: C-ABI calls (one launch '
                               'each, except: strided input gradients = one per parity class, weight gradients = kernel + slab '
                               'reduce); the rocprofv3 kernel count per step is in profiles/.  With the recorded step the host makes a '
                               'handful of calls per step (stream joins, asm_tape_replay, the optimiser): the launches are the tape\'s'}
    if dp_info is not None:
      out['dp'] = dp_info
    if recipe is not None:
      out['recipe'] = recipe
    if args.rehearsal_one_gpu:
      out['data'] += '; REHEARSAL: %d ranks sharing ONE GPU over gloo -- exercises the N > 1 code path, the value is not a scaling number' % world
    if dry:
      out['data'] = 'DRY RUN on CPU (test double of the C ABI, gloo): control flow only, the value means nothing'
    if single is not None:
      out['single_stream'] = single
    if world == 1 and not args.no_cpu_baseline:
      try:
        out['cpu_baseline'] = cpu_baseline(args.workload)
      except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
        out['cpu_baseline'] = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': 'failed: %r' % (e,)}
    _flush_c_stdio()
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
