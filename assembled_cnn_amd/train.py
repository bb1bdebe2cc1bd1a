"""One training step of the Assemble-ResNet path on MI355X: the body of the reference's
``resnet_model_fn`` (nets/run_loop_classification.py:60-234) + ``get_train_op``
(nets/optimizer_setting.py:23-38), plus the learning-rate schedule (functions/model_fns.py:36-95).

  raw images -> [mixup] + mean-subtract + cast (one fused kernel)        utils/data_util.py:97-158,
                                                                          preprocessing/imagenet_preprocessing.py:122-155
  -> Model forward (HIP) -> softmax-CE (+label smoothing) + KD, fused fwd/bwd   losses/cls_losses.py:23-41,
                                                                          run_loop_classification.py:156-162
  -> hand-written backward tape -> [gradient all-reduce over RCCL]       official/utils/misc/distribution_utils.py:24-45
  -> momentum-SGD with the L2 term folded in (one flat launch per decay group)  run_loop_classification.py:166-178
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import nn, ops
from .model import Model

IMAGENET_NUM_CLASSES = 1001          # functions/data_config.py:44
IMAGENET_NUM_TRAIN_IMAGES = 1281167  # functions/data_config.py:45-47


@dataclass
class HParams(object):
  """The flag surface of nets/hparams_config.py (+ official/utils/flags) that reaches the hot path.
  Names and defaults follow the reference."""
  # architecture
  resnet_size: int = 50                       # main_classification.py:35-36
  resnet_version: int = 1                     # nets/hparams_config.py:170  (2 == BigLittleNet)
  use_resnet_d: bool = False                  # :150
  use_se_block: bool = False                  # :154
  use_sk_block: bool = False                  # :158
  anti_alias_filter_size: int = 0             # :161
  anti_alias_type: str = ""                   # :164  substring-matched for 'sconv' / 'proj'
  bl_alpha: int = 2                           # :177
  bl_beta: int = 4                            # :180
  no_downsample: bool = False                 # :120
  pool_type: str = 'gap'                      # :116
  embedding_size: int = 0                     # :77
  zero_gamma: bool = False                    # :214
  bn_momentum: float = 0.997                  # :73
  num_classes: int = IMAGENET_NUM_CLASSES
  # regularisation / loss
  label_smoothing: float = 0.0                # :201
  kd_temp: float = 0.0                        # :204
  mixup_type: int = 0                         # :137
  weight_decay: float = 4e-5                  # :188
  use_dropblock: bool = False                 # :192
  dropblock_kp: List[float] = field(default_factory=lambda: [1.0, 0.9])
  cls_loss_type: str = 'softmax'              # :98
  # optimisation
  base_learning_rate: float = 0.01            # :56
  learning_rate_decay_type: str = 'exponential'  # :64
  lr_warmup_epochs: int = 0                   # :212
  momentum: float = 0.9                       # :69
  num_epochs_per_decay: float = 2.0           # :48
  learning_rate_decay_factor: float = 0.94    # :52
  end_learning_rate: float = 1e-4             # :60
  piecewise_lr_boundary_epochs: List[int] = field(default_factory=lambda: [30, 60, 80, 90])
  piecewise_lr_decay_rates: List[float] = field(default_factory=lambda: [1, 0.1, 0.01, 0.001, 1e-4])
  train_epochs: int = 90                      # main_classification.py:38
  batch_size: int = 32                        # global batch, official/utils/flags/_base.py:91
  dtype: str = 'bf16'                         # reference: fp16|fp32 (_performance.py:29-42); bf16 here
  loss_scale: Optional[float] = None          # fp16 -> 128, else 1 (_performance.py:39-42)
  num_images_train: int = IMAGENET_NUM_TRAIN_IMAGES

  def get_loss_scale(self) -> float:
    if self.loss_scale is not None:
      return float(self.loss_scale)
    return 128.0 if self.dtype == 'fp16' else 1.0

  def to_cfg(self):
    """The flags as the C ABI's struct asm_model_cfg (include/asm_hip.h) -- what asm_model_plan consumes."""
    from . import lib
    aa = (lib.ASM_AA_SCONV if 'sconv' in self.anti_alias_type else 0) | (lib.ASM_AA_PROJ if 'proj' in self.anti_alias_type else 0)
    if self.pool_type not in lib.POOL_TYPES:
      raise NotImplementedError(self.pool_type)
    dt = {'bf16': lib.ASM_BF16, 'fp16': lib.ASM_F16, 'fp32': lib.ASM_F32}[self.dtype]
    return lib.ModelCfg(resnet_size=self.resnet_size, resnet_version=self.resnet_version, num_classes=self.num_classes,
                        use_se_block=int(self.use_se_block), use_sk_block=int(self.use_sk_block),
                        use_resnet_d=int(self.use_resnet_d), anti_alias_filter_size=self.anti_alias_filter_size,
                        anti_alias_type=aa, bl_alpha=self.bl_alpha, bl_beta=self.bl_beta, zero_gamma=int(self.zero_gamma),
                        no_downsample=int(self.no_downsample), pool_type=lib.POOL_TYPES[self.pool_type],
                        embedding_size=self.embedding_size, dtype=dt, mixup_type=self.mixup_type,
                        bn_momentum=self.bn_momentum, bn_eps=1e-5, loss_scale=self.get_loss_scale(),
                        label_smoothing=self.label_smoothing, kd_temp=self.kd_temp, weight_decay=self.weight_decay,
                        momentum=self.momentum)

  def make_model(self, seed=0, device='cuda') -> Model:
    if self.resnet_size < 50:  # functions/model_fns.py:202-206
      assert not (self.use_dropblock or self.use_se_block or self.use_sk_block or self.use_resnet_d)
    return Model(self.resnet_size, None, num_classes=self.num_classes, resnet_version=self.resnet_version,
                 dtype=self.dtype, no_downsample=self.no_downsample, zero_gamma=self.zero_gamma,
                 use_se_block=self.use_se_block, use_sk_block=self.use_sk_block, bn_momentum=self.bn_momentum,
                 embedding_size=self.embedding_size, anti_alias_filter_size=self.anti_alias_filter_size,
                 anti_alias_type=self.anti_alias_type, pool_type=self.pool_type, loss_type=self.cls_loss_type,
                 bl_alpha=self.bl_alpha, bl_beta=self.bl_beta, seed=seed, device=device)


def learning_rate_with_decay(learning_rate_decay_type, batch_size, batch_denom, num_images, num_epochs_per_decay,
                             learning_rate_decay_factor, end_learning_rate, piecewise_lr_boundary_epochs,
                             piecewise_lr_decay_rates, base_lr, warmup_epochs=0, train_epochs=None
                             ) -> Callable[[int], float]:
  """functions/model_fns.py:36-95 as a host function of the global step."""
  initial_learning_rate = base_lr * batch_size / batch_denom
  batches_per_epoch = num_images / batch_size
  decay_steps = int(batches_per_epoch * num_epochs_per_decay)
  kind = learning_rate_decay_type
  if kind not in ('exponential', 'fixed', 'polynomial', 'piecewise', 'cosine'):
    raise NotImplementedError(kind)

  def learning_rate_fn(global_step: int) -> float:
    warmup_steps = int(batches_per_epoch * warmup_epochs)
    if warmup_steps > 0 and global_step < warmup_steps:
      return initial_learning_rate * float(global_step) / float(warmup_steps)
    step = global_step - warmup_steps
    if kind == 'exponential':
      return initial_learning_rate * learning_rate_decay_factor ** (step // decay_steps)
    if kind == 'fixed':
      return base_lr
    if kind == 'polynomial':
      s = min(step, decay_steps)
      return (initial_learning_rate - end_learning_rate) * (1.0 - s / decay_steps) + end_learning_rate
    if kind == 'piecewise':
      boundaries = [int(batches_per_epoch * e) for e in piecewise_lr_boundary_epochs]
      values = [initial_learning_rate * float(d) for d in piecewise_lr_decay_rates]
      for b, v in zip(boundaries, values):
        if global_step <= b:
          return v
      return values[-1]
    total_batches = int(batches_per_epoch * train_epochs) - warmup_steps
    s = min(step, total_batches)
    return 0.5 * (1.0 + math.cos(math.pi * s / total_batches)) * initial_learning_rate
  return learning_rate_fn


def keep_prob_decay(starter_kp: float, end_kp: float, decay_steps: int) -> Callable[[int], float]:
  """functions/model_fns.py:26-33: tf.train.polynomial_decay(power=1, cycle=False) of the DropBlock keep_prob."""
  def keep_prob_decay_fn(global_step: int) -> float:
    s = min(global_step, decay_steps)
    return (starter_kp - end_kp) * (1.0 - s / decay_steps) + end_kp
  return keep_prob_decay_fn


def keep_prob_fn_from_hparams(p: "HParams") -> Optional[Callable[[int], float]]:
  """model_fn_cls wiring (functions/model_fns.py:221-226)."""
  if not p.use_dropblock:
    return None
  batches_per_epoch = p.num_images_train / p.batch_size
  return keep_prob_decay(p.dropblock_kp[0], p.dropblock_kp[1], int(p.train_epochs * batches_per_epoch))


def lr_fn_from_hparams(p: HParams) -> Callable[[int], float]:
  """model_fn_cls wiring (functions/model_fns.py:208-219): batch_denom == batch_size."""
  return learning_rate_with_decay(p.learning_rate_decay_type, p.batch_size, p.batch_size, p.num_images_train,
                                  p.num_epochs_per_decay, p.learning_rate_decay_factor, p.end_learning_rate,
                                  p.piecewise_lr_boundary_epochs, p.piecewise_lr_decay_rates, p.base_learning_rate,
                                  warmup_epochs=p.lr_warmup_epochs, train_epochs=p.train_epochs)


class Trainer(object):
  """Holds the model + optimiser state and runs training steps.  ``grad_sync`` (see dp.py) is called
  between backward and the optimiser with the flat fp32 gradient arena."""

  AUTO_WARMUP = 3      # eager steps of one input signature before train_step records itself
  tape_check_skipped = None   # why the last capture could not compare its tape with the captured graph (None: it was compared)

  def __init__(self, hparams: HParams, seed: int = 0, device='cuda', grad_sync=None, world_size: int = 1,
               recorded: Optional[bool] = None):
    """``recorded``: None (default) -- on a GPU, train_step records itself after AUTO_WARMUP eager steps with inputs of one
    shape (Trainer.capture: the launch tape) and replays from then on, re-records when the shapes change and falls back
    to the eager step, with a warning, if recording fails; True -- the same, but a failed recording raises; False --
    never on its own (capture() can still be called).  ASM_STEP_TAPE=0 switches the automatic recording off."""
    self.p = hparams
    self.model = hparams.make_model(seed, device)
    self.lr_fn = lr_fn_from_hparams(hparams)
    self.keep_prob_fn = keep_prob_fn_from_hparams(hparams)
    self.global_step = 0
    self.eval_state = None
    self.grad_sync = grad_sync
    self.world_size = world_size
    self.last = {}
    self._graph = self._static = self._graph_out = None
    self._tape = self._cap_stream = None
    # the stream a captured step is recorded and replayed on: taken from the pool BEFORE the model takes its side streams
    dev = torch.device(device)
    self._step_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
    self.recorded = recorded
    self._auto = dev.type == 'cuda' and recorded is not False and ops.knob('ASM_STEP_TAPE', '1') != '0'
    self._auto_sig, self._auto_n, self._auto_made = None, 0, False
    self._capturing = False
    self._baked = None
    self.step_mode = 'eager'                 # 'eager' | 'recorded' | 'eager (recording failed: ...)'
    # DropBlock with its draws and gamma in static device buffers (nn.DropBlockState): what lets the published recipe
    # (scripts/train_assemble_from_scratch.sh: --use_dropblock=True) run as a recorded step
    # (on the CPU test double only on request: recorded=True, for the host-logic tests)
    self._db = nn.DropBlockState(dev) if (self.keep_prob_fn is not None and recorded is not False and
                                          (dev.type == 'cuda' or recorded is True)) else None

  @property
  def stream(self):
    """The HIP stream captured steps are recorded and replayed on (None on the CPU test double).  A training loop that
    runs under ``with torch.cuda.stream(trainer.stream)`` keeps the default stream out of the step entirely, which is
    the fastest arrangement measured (see _replay)."""
    return self._step_stream

  # -----------------------------------------------------------------------------------------------
  def set_streams(self, on: bool, fresh: bool = False):
    """Side streams on (the default: weight gradients on two streams beside the input-gradient chain, the big branch of
    every BigLittle stage beside the little one) or off (every kernel on the compute stream).  Results are bit-identical
    either way (tests/test_gpu_model.py); only the overlap changes.  Switching off parks the stream objects and switching
    on again reuses them; ``fresh`` creates new ones instead.  The setting belongs to this trainer's model (the process-wide
    ASM_WGRAD_STREAM / ASM_BL_STREAMS switches are only the default) and cannot change under a recorded step."""
    if self._graph is not None:
      if not self._auto_made:
        raise RuntimeError('the stream setting is frozen into the captured step: release_graph() first')
      self.release_graph()                 # the trainer's own recording: drop it, the next steps record again
    self.model.side_streams = bool(on)     # this model only: no other Trainer / Model of the process changes
    a = self.model.arena
    if a.finalized:
      if on:
        if fresh:
          a.disable_side_stream()
          old = self.model._bl_stream
          if old is not None:
            ops.stream_join(torch.cuda.current_stream(), old)
            a.extra_streams = [s_ for s_ in a.extra_streams if s_ is not old]
          self.model._bl_stream = None
        a.enable_side_stream(fresh=fresh)
      else:
        a.disable_side_stream()

  def calibrate_streams(self, step_fn, steps: int = 3, margin: float = 0.03, redraws: int = 0):
    """Time ``steps`` EAGER training steps with the side streams on and off and keep the faster setting.

    Why this exists (round 4): the side streams are worth ~1 ms per step, but they also make the eager step more expensive
    to ENQUEUE (an event record + a stream wait per weight gradient), and the eager step has little host head-room: ~14.5 ms
    of Python per step against ~25 ms of GPU time.  On a busy host the step becomes host-bound, and the side-stream form
    degrades first: with one busy loop beside the Python thread 30.2 ms per step against 26.7 on one stream, with two 48.5
    against 33.3 (tools/debug/host_contention.sh; the GPU boxes this was developed on are shared, 256 hardware threads at
    load averages of 40 - 65).  Rounds 3 - 4 first read those runs as a problem of where the runtime places streams on the
    hardware queues; ``redraws`` (fresh stream objects when the side streams lose) is what is left of that reading and is
    off by default.  The measurement itself stays: an eager trainer keeps the side streams unless they are more than
    ``margin`` slower than one stream in THIS process on THIS host.  The real remedy for a busy host is to take the host
    out of the step -- Trainer.capture replays the recorded launches in ~3 ms of host time per step and needs no
    calibration.  ``step_fn()`` runs one real training step (the steps taken here are ordinary steps).  Returns the
    measurements."""
    import time

    def timed(on, fresh=False):
      self.set_streams(on, fresh=fresh)
      step_fn()                               # the first step after a switch warms the allocator on the new streams
      torch.cuda.synchronize()
      t0 = time.time()
      for _ in range(steps):
        step_fn()
      torch.cuda.synchronize()
      return round(1000.0 * (time.time() - t0) / steps, 3)

    # The steps timed here must be EAGER ones: a trainer that records itself (the default on a GPU) would reach AUTO_WARMUP
    # inside the timed window and time a capture instead.  Self-recording is suspended for the duration and
    # the trainer's own recording, if any, dropped; a caller-made capture() is frozen with its stream setting (set_streams
    # raises, as before).
    auto = self._auto
    self._auto = False
    if self._graph is not None and self._auto_made:
      self.release_graph()
    try:
      res = {'side_streams_ms': timed(True), 'single_stream_ms': timed(False), 'redraws': 0}
      keep = res['side_streams_ms'] <= res['single_stream_ms'] * (1.0 + margin)
      while not keep and res['redraws'] < redraws:
        res['redraws'] += 1
        res['side_streams_ms'] = timed(True, fresh=True)
        keep = res['side_streams_ms'] <= res['single_stream_ms'] * (1.0 + margin)
      self.set_streams(keep)
    finally:
      self._auto = auto
      self._auto_n = 0
    if self._graph is not None:
      raise RuntimeError('calibrate_streams: a step was recorded while eager steps were being timed')
    res['chosen'] = 'side streams' if keep else 'single stream'
    return res

  # -----------------------------------------------------------------------------------------------
  def split_labels(self, labels):
    """labels -> (hard targets, teacher logits or None).  labels: int [Bin] or, with kd_temp > 0, float32 [Bin, 2C] =
    concat(one-hot, teacher logits) (run_loop_classification.py:90-96).  Framework slices / casts: this is the part of the
    input side that stays OUTSIDE a recorded step (capture keeps the two halves as separate static buffers)."""
    p = self.p
    C = p.num_classes
    if p.kd_temp > 0:
      if labels.dim() != 2 or labels.shape[1] != 2 * C:
        raise ValueError('kd_temp > 0 expects labels [B, 2*num_classes]')
      return labels[:, :C].contiguous(), labels[:, C:].contiguous()
    return labels.to(torch.int32).contiguous(), None

  def prepare_inputs(self, images, labels, lam1=None, lam2=None):
    """images: [Bin,H,W,3] uint8 / float32 (0..255).  labels: int32 [Bin] or, with kd_temp > 0,
    float32 [Bin, 2C] = concat(one-hot, teacher logits) (run_loop_classification.py:90-96).
    Returns (stem input halo buffer, dense targets, teacher probabilities or None)."""
    hard, tlogits = self.split_labels(labels)
    return self._prepare(images, hard, tlogits, lam1, lam2)

  def _prepare(self, images, hard, tlogits, lam1, lam2):
    """The input side of a step from its split labels: library launches only (a recorded step contains it)."""
    p = self.p
    C = p.num_classes
    Bin = images.shape[0]
    if p.kd_temp > 0:
      onehot = hard
      teacher = ops.softmax_rows(tlogits, Bin, C, 1.0 / p.kd_temp)
    else:
      onehot = ops.onehot(hard, Bin, C)
      teacher = None
    mt = p.mixup_type
    if mt not in (0, 1, 2):
      raise ValueError('mixup_type must be 0, 1 or 2')
    if mt and lam1 is None:
      raise ValueError('mixup needs the Beta(0.2, 0.2) draws (lam1 [, lam2])')
    x = ops.mixup_meansub(images if images.is_contiguous() else images.contiguous(), mt, lam1, lam2)
    if mt:
      onehot_src = onehot
      onehot = ops.mixup_labels(onehot, mt, lam1, lam2)
      if teacher is not None:
        if mt == 2:
          # the reference mixes the second-half teacher targets from the HARD labels y1, not y1_t
          # (utils/data_util.py:154).  Reproduced: first half = type-1 mix of the teacher; second half = the
          # type-2 mix of [y1 ; y2_t], whose second half is lam2*y1 + (1-lam2)*reverse(y2_t).  Row ranges are moved
          # with library copies, so that a recorded step sees them.
          half = Bin // 2
          first = ops.mixup_labels(teacher, 1, lam1, None)
          src2 = ops.empty((Bin, C), torch.float32, teacher)
          ops.memcpy(src2[:half], onehot_src[:half])
          ops.memcpy(src2[half:], teacher[half:])
          second = ops.mixup_labels(src2, 2, lam1, lam2)
          teacher = ops.empty((Bin, C), torch.float32, second)
          ops.memcpy(teacher[:half], first)
          ops.memcpy(teacher[half:], second[half:])
        else:
          teacher = ops.mixup_labels(teacher, mt, lam1, lam2)
    return x, onehot, teacher

  def sample_mixup_lambdas(self, n: int, alpha: float = 0.2, rng=None):
    """Beta(alpha, alpha) draws on the host (utils/data_util.py:98,105), uploaded as float32."""
    import numpy as np
    rng = rng if rng is not None else np.random.default_rng()
    lam = rng.beta(alpha, alpha, size=n).astype(np.float32)
    return torch.from_numpy(lam).to(self.model.device)

  # -----------------------------------------------------------------------------------------------
  def train_step(self, images, labels, lam1=None, lam2=None, lr: Optional[float] = None,
                 dropblock_uniforms=None):
    """One optimisation step.  A recorded step (capture(), or the trainer's own recording after AUTO_WARMUP eager steps) is
    replayed; ``dropblock_uniforms`` (tests: the draws of every DropBlock call in creation order) go into the static
    draw buffers of a recorded step, or straight to the layers of an eager one."""
    if self._graph is not None and self._baked_state() != self._baked:
      if not self._auto_made:
        raise RuntimeError('grad_sync or a by-value hyper-parameter (loss scale, label smoothing, KD temperature, weight '
                           'decay, momentum, mixup type) changed after capture(): release_graph() and capture again')
      self.release_graph()                 # the trainer's own recording: drop it, the eager steps below record again
    if self._graph is not None:
      if self._auto_made and self._signature(images, labels, lam1, lam2) != self._auto_sig:
        self.release_graph()               # other shapes: back to the eager step, which records itself again
      elif dropblock_uniforms is not None and self._db is None:
        pass                               # a step captured without DropBlock state: run this one eagerly
      else:
        return self._replay(images, labels, lam1, lam2, lr, dropblock_uniforms)
    loss_rows, loss_scale, keep_prob = self._forward_backward(images, labels, lam1, lam2, dropblock_uniforms)
    out = self._apply(loss_rows, loss_scale, keep_prob, lr)
    if self._auto and self._graph is None and not self._capturing:
      self._auto_step(images, labels, lam1, lam2)
    return out

  @staticmethod
  def _signature(images, labels, lam1, lam2):
    return tuple((tuple(t.shape), t.dtype if t.dim() != 1 or t.dtype.is_floating_point else torch.int32, t.device)
                 if t is not None else None for t in (images, labels, lam1, lam2))

  def _auto_step(self, images, labels, lam1, lam2):
    """after an eager step: count steps of one input signature; record once there were AUTO_WARMUP of them"""
    if not images.is_cuda or (self.grad_sync is not None and not hasattr(self.grad_sync, 'launch_recorded')):
      return
    sig = self._signature(images, labels, lam1, lam2)
    if sig != self._auto_sig:
      self._auto_sig, self._auto_n = sig, 0
    self._auto_n += 1
    if self._auto_n < self.AUTO_WARMUP:
      return
    try:
      self.capture(images, labels, lam1, lam2, warmup=0)
      self._auto_made = True
    except Exception as e:                 # loud, once: the trainer stays eager
      if self.recorded is True:
        raise
      import warnings
      self._auto = False
      self.step_mode = 'eager (recording failed: %s)' % (str(e).splitlines()[0][:200] if str(e) else type(e).__name__)
      warnings.warn('assembled_cnn_amd.Trainer: recording the training step failed, staying with the eager step '
                    '(14 ms of host time per step instead of 3): %r' % (e,), RuntimeWarning)

  def _forward_backward(self, images, labels, lam1, lam2, dropblock_uniforms=None):
    """Everything of a step that does not depend on the step number: inputs -> loss rows, gradients in the arena."""
    hard, tlogits = self.split_labels(labels)
    return self._forward_backward_split(images, hard, tlogits, lam1, lam2, dropblock_uniforms)

  def _forward_backward_split(self, images, hard, tlogits, lam1, lam2, dropblock_uniforms=None, db_prepared=False):
    """_forward_backward behind split_labels: library launches only (this is what a recording holds).  With DropBlock state (self._db) the draws and gamma of this step are written
    into its static buffers first -- unless the caller has done that already (db_prepared: a step being recorded)."""
    p = self.p
    m = self.model
    keep_prob = self.keep_prob_fn(self.global_step) if self.keep_prob_fn else 1.0
    db = self._db
    if db is not None and not db_prepared:
      db.begin(keep_prob, dropblock_uniforms, self._db_rng())
    elif db is not None:
      db._i = 0
    x, onehot, teacher = self._prepare(images, hard, tlogits, lam1, lam2)
    B = x.shape[0]
    m(x, True, use_resnet_d=p.use_resnet_d, prepadded=True, keep_prob=keep_prob,
      dropblock_uniforms=dropblock_uniforms if db is None else None, db_static=db)
    if db is not None:
      db.end()
    loss_scale = p.get_loss_scale()
    if p.cls_loss_type == 'softmax':      # losses/cls_losses.py:27-33 (+ KD, run_loop_classification.py:156-162)
      loss_rows, dlogits = ops.softmax_ce(m.logits_padded, m.ldc, onehot, teacher, B, p.num_classes,
                                          p.label_smoothing, p.kd_temp, loss_scale, m.ldc)
    elif p.cls_loss_type == 'sigmoid':    # losses/cls_losses.py:34-38
      if teacher is not None:
        raise NotImplementedError('KD on top of the sigmoid loss is not implemented on the HIP path')
      sig, dlogits = ops.sigmoid_ce(m.logits_padded, m.ldc, onehot, B, p.num_classes, loss_scale, m.ldc)
      loss_rows = sig[:1]
    else:
      raise AssertionError('cross_entropy is None')   # losses/cls_losses.py:40
    m.backward(dlogits)
    return loss_rows, loss_scale, keep_prob

  def _db_rng(self):
    """the generator DropBlock draws come from when the caller supplies none (the model's, so that an eager trainer and a
    recorded one with the same seed draw the same sequence)"""
    m = self.model
    if m._db_rng is None:
      m._db_rng = torch.Generator(device=m.device)
      m._db_rng.manual_seed(m.seed + 12345)
    return m._db_rng

  def _apply(self, loss_rows, loss_scale, keep_prob, lr=None):
    """[gradient exchange ->] momentum-SGD at the step's learning rate (a host scalar: outside any captured graph)."""
    p = self.p
    a = self.model.arena
    if self.grad_sync is not None:
      self.grad_sync(a.g32)
    lr = self.lr_fn(self.global_step) if lr is None else lr
    gs = 1.0 / (loss_scale * self.world_size)   # un-scale the loss scale; SUM-all-reduce -> mean over replicas
    nd = a.decay_elems
    if nd:
      ops.sgd_momentum(a.w32[:nd], a.m32[:nd], a.g32[:nd], a.w16[:nd], lr, p.momentum, p.weight_decay, gs)
    if a.total_elems > nd:
      ops.sgd_momentum(a.w32[nd:], a.m32[nd:], a.g32[nd:], a.w16[nd:], lr, p.momentum, 0.0, gs)
    a.refresh_derived()
    self.global_step += 1
    self.last = {'loss_rows': loss_rows, 'lr': lr, 'keep_prob': keep_prob}
    return loss_rows

  # ---- the step as a recorded launch sequence ---------------------------------------------------------------------------
  def capture(self, images, labels, lam1=None, lam2=None, warmup: int = 2, capture_error_mode: str = 'global',
              replay: str = 'tape'):
    """Record inputs -> forward -> loss -> backward (every launch of it, side streams included) over static input buffers;
    train_step then copies its arguments into those buffers, replays the recording with ONE host call instead of ~930
    launches' worth of Python, and runs the exchange + optimiser as usual (their learning rate is a host scalar).

    Why: the eager step enqueues in 14 - 15 ms of host time against 25 ms of GPU time.  On a busy host -- the GPU boxes this
    was developed on are shared: 256 hardware threads, load averages of 40 - 65 -- that head-room goes: with one competitor
    on the Python thread's core the step is host-bound at 30 ms, with two at 48 (tools/debug/host_contention.sh).

    ``replay='tape'`` (default): the library writes the launches down while they are captured (csrc/tape.hip) and
    ``asm_tape_replay`` issues them again through hipLaunchKernel on the recorded streams: ~2 - 3 ms of host time per step.
    The HIP graph of the capture is never launched; it is kept for its private memory pool, which is what holds every
    buffer of the step at its recorded address (and never hands a block to another stream).
    ``replay='graph'``: hipGraphLaunch of the captured graph; ROCm 7 spends 11.5 - 22 ms of host time per launch on the
    step's nodes, so this only pays where the eager step is host-bound anyway.

    Both replays are bit-identical to the eager step (tests/test_gpu_model.py).  DropBlock (the published recipe,
    scripts/train_assemble_from_scratch.sh:22) is recorded with its draws and its gamma in static device buffers that
    every replay rewrites first (nn.DropBlockState).  KD labels are split into their two halves OUTSIDE the recording
    (split_labels: framework slices) and everything behind the split is library launches.  The recording is checked
    against the captured HIP graph: the graph must hold exactly the kernels and copies the tape wrote down -- a framework
    kernel that slipped into the recorded region would be in the graph only, i.e. silently missing from every replay.
    With a gradient exchange attached (dp.GradSync) only the tape works: the
    bucket launches the host interleaves with the backward pass become segment boundaries of the tape, and the replayed
    step hands bucket k to RCCL after segment k (tests/test_gpu_dp_rccl.py).  ``warmup`` real training steps run first
    (one-time initialisation must not be recorded).
    Every activation of a step stays allocated until release_graph(); ``capture_error_mode='thread_local'`` if other
    threads (an input pipeline) touch the device while capturing.  ASM_* switches are frozen into the recording."""
    if self._graph is not None:
      raise RuntimeError('a step is already captured: release_graph() first')
    if replay not in ('tape', 'graph'):
      raise ValueError("replay must be 'tape' or 'graph'")
    if self.keep_prob_fn is not None and self._db is None:
      raise NotImplementedError('DropBlock changes keep_prob and its draws every step: recording needs the static DropBlock '
                                'buffers (a Trainer made with recorded=False has none)')
    if self.grad_sync is not None and replay != 'tape':
      raise NotImplementedError('with a gradient exchange attached the bucket launches are host-driven: a HIP graph cannot '
                                'hold them; the launch tape is cut into segments at the bucket launches instead')
    if self.grad_sync is not None and not hasattr(self.grad_sync, 'launch_recorded'):
      raise NotImplementedError('capture needs a dp.GradSync as the gradient exchange (segmented replay)')
    if not images.is_cuda:
      raise RuntimeError('capture needs device tensors')
    hard, tlogits = self.split_labels(labels)    # framework kernels (cast / column slices): not part of the recorded step
    static = [t.clone() if t is not None else None for t in (images, hard, tlogits, lam1, lam2)]
    sig = self._signature(images, labels, lam1, lam2)
    cur = torch.cuda.current_stream(images.device)
    if cur != torch.cuda.default_stream(images.device):
      cap = cur           # already on a stream of the caller's: record there (a capture cannot run on the default stream)
    elif self._step_stream is not None:
      cap = self._step_stream
    else:
      cap = torch.cuda.Stream(device=images.device)
    if cap != cur:
      ops.stream_join(cap, cur)
    with torch.cuda.stream(cap):
      self._capturing = True      # the warm-up steps below must not start a recording of their own
      try:
        for _ in range(warmup):   # one-time initialisation (workspaces, kernel attributes, side streams) stays out of the recording
          self.train_step(images, labels, lam1, lam2)
      finally:
        self._capturing = False
      if self._db is not None:
        if not self._db.known:
          raise RuntimeError('capture with DropBlock needs one eager step first (warmup >= 1): it discovers the DropBlock calls')
    torch.cuda.synchronize()
    # tape replay never launches the captured graph: it is not even instantiated (keep_graph), only kept for its memory pool
    g = torch.cuda.CUDAGraph(keep_graph=(replay == 'tape'))
    tape = None
    gs = self.grad_sync
    if gs is not None and capture_error_mode == 'global':
      capture_error_mode = 'thread_local'      # the collective library's watchdog thread queries events while we capture
    try:
      with torch.cuda.graph(g, stream=cap, capture_error_mode=capture_error_mode):
        if replay == 'tape':
          tape = ops.tape_begin()
        if gs is not None:
          gs.begin_recording()      # bucket launches of the recorded backward pass become segment boundaries of the tape
        try:
          out = self._forward_backward_split(*static, db_prepared=True)
        finally:
          if gs is not None:
            gs.end_recording()
          if tape is not None:
            ops.tape_end()
      if gs is not None and ops.tape_info(tape)['segments'] != len(gs.recorded) + 1:
        raise RuntimeError('tape segments and recorded bucket launches disagree')
      if tape is not None:
        self._check_tape_against_graph(g, tape)
    except BaseException:
      if tape is not None:        # a recording that failed half-way is dropped, the trainer stays eager
        ops.tape_free(tape)
      raise
    self._graph, self._static, self._graph_out = g, static, out
    self._tape, self._cap_stream = tape, cap
    self._auto_sig, self._auto_made = sig, False
    self._baked = self._baked_state()
    self.step_mode = 'recorded'
    return self

  @staticmethod
  def _check_tape_against_graph(g, tape):
    """A tape replay issues what the LIBRARY launched while the step was captured.  A framework kernel inside the
    recorded region (a .contiguous(), a cast, a concatenation) is in the captured HIP graph but not on the tape: every
    replay would silently skip it.  Count the graph's kernel and copy / fill nodes through the HIP runtime and compare."""
    info = ops.tape_info(tape)
    try:
      import ctypes
      raw = g.raw_cuda_graph()
      # the runtime torch itself is linked against (already loaded: resolve the symbols from the process image, never
      # dlopen a second libamdhip64 by name -- a system copy would be another runtime instance)
      hip = ctypes.CDLL(None)
      hip.hipGraphGetNodes.restype = ctypes.c_int
      hip.hipGraphNodeGetType.restype = ctypes.c_int
      n = ctypes.c_size_t(0)
      if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
        return None
      nodes = (ctypes.c_void_p * n.value)()
      if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
        return None
      kernels = copies = 0
      for h in nodes[:n.value]:
        ty = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(h), ctypes.byref(ty)) != 0:
          return None
        kernels += ty.value == 0                  # hipGraphNodeTypeKernel
        copies += ty.value in (1, 2)              # hipGraphNodeTypeMemcpy, hipGraphNodeTypeMemset
    except (AttributeError, OSError, RuntimeError) as e:
      Trainer.tape_check_skipped = repr(e)        # this torch / runtime does not expose the raw graph: nothing to compare with
      return None
    # (compared as one total: whether the runtime keeps a device-to-device copy as a copy node or as a blit kernel is its business)
    if kernels + copies != info['launches'] + info['fills']:
      raise RuntimeError('the captured step holds %d kernels and %d copies, the launch tape %d and %d: a framework kernel '
                         'ran inside the recorded region and would be missing from every replay'
                         % (kernels, copies, info['launches'], info['fills']))
    return kernels + copies

  def _baked_state(self):
    """what a recording holds BY VALUE: the gradient exchange it was recorded with and the hyper-parameters that reach the
    kernels as arguments.  A recorded step replayed after one of them changed would silently run with the old value."""
    p = self.p
    return (id(self.grad_sync), float(p.get_loss_scale()), float(p.label_smoothing), float(p.kd_temp), float(p.weight_decay),
            float(p.momentum), int(p.mixup_type))

  def release_graph(self):
    if getattr(self, '_tape', None) is not None:
      torch.cuda.synchronize()
      ops.tape_free(self._tape)
    self._graph = self._static = self._graph_out = None
    self._tape = self._cap_stream = None
    self._auto_made = False
    self._auto_n = 0
    self.step_mode = 'eager'

  def _replay(self, images, labels, lam1, lam2, lr, dropblock_uniforms=None):
    srcs = []
    hard, tlogits = self.split_labels(labels)
    for dst, src in zip(self._static, (images, hard, tlogits, lam1, lam2)):
      if (dst is None) != (src is None) or (dst is not None and (dst.shape != src.shape or dst.dtype != src.dtype)):
        raise ValueError('the captured step takes inputs of the shapes / dtypes it was captured with')
      srcs.append(src)
    # The whole step -- input copies, the recording, the optimiser -- runs on the stream the recording was made on, between
    # two joins with the caller's stream.  Measured (same box, ms per step): everything on that one stream 25.08; recording
    # on it but input copies and optimiser on the default stream 26.2 - 26.5 (the eager step: 25.3 on the default stream,
    # 25.7 - 26.0 on another).
    cur = torch.cuda.current_stream()
    cap = self._cap_stream
    same = cap == cur
    if not same:
      ops.stream_join(cap, cur)
      torch.cuda.set_stream(cap)
    try:
      for dst, src in zip(self._static, srcs):
        if dst is not None and dst.data_ptr() != src.data_ptr():
          dst.copy_(src, non_blocking=True)
      if self._db is not None:      # this step's keep_prob (gamma) and draws into the static DropBlock buffers
        self._db.begin(self.keep_prob_fn(self.global_step), dropblock_uniforms, self._db_rng())
      gs = self.grad_sync
      if self._tape is not None and gs is not None:
        try:
          for k in range(len(gs.recorded)):   # segment k, then the bucket that became ready at its end goes to RCCL
            ops.tape_replay(self._tape, k)
            gs.launch_recorded(k)
          ops.tape_replay(self._tape, len(gs.recorded))
        except BaseException:
          gs.abort()        # half a step's buckets are with the collective library: wait for them, forget the watermark
          raise
      elif self._tape is not None:
        ops.tape_replay(self._tape)       # (the recording ends with its side streams joined into the capture stream)
      else:
        self._graph.replay()
      loss_rows, loss_scale, keep_prob = self._graph_out
      if self.keep_prob_fn is not None:
        keep_prob = self.keep_prob_fn(self.global_step)
      out = self._apply(loss_rows, loss_scale, keep_prob, lr)
    finally:
      if not same:
        torch.cuda.set_stream(cur)
    if not same:
      ops.stream_join(cur, cap)
    return out

  def cross_entropy(self) -> torch.Tensor:
    """mean over the batch of (CE + KD) of the last step (device scalar)."""
    return ops.mean_f32(self.last['loss_rows'])

  def l2_loss(self) -> torch.Tensor:
    """weight_decay * sum(0.5 * w^2) over the decayed set -- reporting only (host-side torch reduce)."""
    a = self.model.arena
    return 0.5 * self.p.weight_decay * (a.w32[:a.decay_elems].double() ** 2).sum()

  # -----------------------------------------------------------------------------------------------
  def eval_logits(self, images_meansub_nhwc: torch.Tensor) -> torch.Tensor:
    """Inference forward (BN moving statistics), logits float32 [B, C]."""
    return self.model(images_meansub_nhwc, False, use_resnet_d=self.p.use_resnet_d)

  # ---- evaluation metrics on device (nets/run_loop_classification.py:208-219) ----------------------
  def eval_reset(self):
    self.eval_state = torch.zeros((33,), dtype=torch.float32, device=self.model.device)

  def eval_step(self, images_meansub_nhwc: torch.Tensor, labels: torch.Tensor):
    """One evaluation batch: forward with moving statistics, then accuracy / top-5 / ECE-bin accumulation
    on the device.  Returns the predicted classes (int32)."""
    if self.eval_state is None:
      self.eval_reset()
    m = self.model
    self.eval_logits(images_meansub_nhwc)
    B = images_meansub_nhwc.shape[0]
    pred, conf, top1, top5 = ops.eval_rows(m.logits_padded, m.ldc, labels.to(torch.int32).contiguous(), B,
                                           self.p.num_classes)
    ops.eval_accumulate(conf, top1, top5, self.eval_state)
    return pred

  def eval_result(self, reduce: bool = True) -> dict:
    """{'accuracy', 'accuracy_top_5', 'ece'} from the running state (metric/ece_metric.py:271-279).  With an
    initialised process group the 33 running sums are all-reduced first (cross-replica aggregation, :281-298), so
    every rank reports the metrics of the whole evaluation set."""
    st = self.eval_state.clone()
    if reduce:
      import torch.distributed as dist
      if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
    st = st.double().cpu()
    n = float(st[2])
    correct, conf, cnt = st[3:13], st[13:23], st[23:33]
    eps = 1e-7
    ece = float(((cnt / cnt.sum()) * ((correct / (eps + cnt)) - (conf / (eps + cnt))).abs()).sum()) if n else 0.0
    return {'accuracy': float(st[0]) / n if n else 0.0, 'accuracy_top_5': float(st[1]) / n if n else 0.0, 'ece': ece,
            'count': n}
