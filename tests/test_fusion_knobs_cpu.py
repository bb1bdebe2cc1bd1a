"""Host-side fusion plumbing (CPU test double of the C ABI): every lazy / fused path of nn.py -- lazily masked shortcut
gradients, the deferred + dual batch norm of a projection shortcut, the pooled gradient gathered by conv1's input
gradient, the batch-norm backward sums reduced in the producing input gradient's epilogue, the masked block sum at a BigLittle merge, the reordered tape
of a projection block -- must give the gradients of the plain one-kernel-per-op path.  The double computes both in fp32
with bf16 rounding where the kernels store bf16, so the two differ by rounding only."""
import pytest
import torch

from tests import model_parity as MP
from tests import util


def _run(monkeypatch, fused: bool, name='a-r50-d', batch=2, size=64, gatherable=True):
  from assembled_cnn_amd import nn, ops
  if not gatherable:      # the pooled gradient stays pending until somebody reads .grad, which scatters it
    monkeypatch.setattr(ops, 'dgrad_pool_ok', lambda d: False)
  for k in ('ASM_POOL_FUSE', 'ASM_BN_DUAL', 'ASM_DENSE_SMALL', 'ASM_SK_FUSED', 'ASM_BN_RED'):
    util.set_knob(monkeypatch, k, '1' if fused else '0')
  monkeypatch.setattr(nn, 'DEFER_BN', fused)
  monkeypatch.setattr(nn, 'LAZY_DZ', fused)
  _, pm = MP.make_pair(name, 'cpu', batch, size)
  _, x, _ = MP.inputs(batch, size)
  lp = pm(x, True, use_resnet_d=MP.uses_d(name))
  dl = torch.zeros((batch, 1, 1, pm.ldc), dtype=torch.bfloat16)
  dl[:, 0, 0, :1001] = (torch.softmax(lp.float(), 1) / batch).to(torch.bfloat16)
  pm.backward(dl)
  grads = {n: pm.arena.g(n).float().clone() for n in pm.arena.specs}
  return lp.float().clone(), grads


@pytest.mark.parametrize('name', ['a-r50-d'])
def test_fused_and_plain_paths_give_the_same_gradients(cpu_double, monkeypatch, name):
  lf, gf = _run(monkeypatch, True, name)
  lp, gp = _run(monkeypatch, False, name)
  assert float((lf - lp).norm() / lp.norm()) <= 2e-2
  fa = torch.cat([g.reshape(-1) for g in gf.values()])
  fb = torch.cat([g.reshape(-1) for g in gp.values()])
  cos = float((fa * fb).sum() / (fa.norm() * fb.norm()))
  assert cos >= 0.97, 'global gradient cosine %.4f between the fused and the plain host paths' % cos
  worst = min(float((gf[n] * gp[n]).sum() / (gf[n].norm() * gp[n].norm() + 1e-30)) for n in gf if float(gp[n].norm()) > 0)
  assert worst >= 0.7, 'worst per-variable gradient cosine %.3f' % worst
  # the same with no convolution able to gather the pooled gradient: the fallback scatter must give the same sum
  _, gn = _run(monkeypatch, True, name, gatherable=False)
  fc = torch.cat([g.reshape(-1) for g in gn.values()])
  assert float((fa * fc).sum() / (fa.norm() * fc.norm())) >= 0.97


def test_projection_block_keeps_the_gradient_watermark_monotone(cpu_double):
  """The shortcut branch of a projection block runs its backward BEFORE the main branch, yet the gradient-ready
  notifications must still arrive in reverse creation order per segment (what dp.GradSync.notify enforces)."""
  _, pm = MP.make_pair('a-r50-d', 'cpu', 2, 64)
  _, x, _ = MP.inputs(2, 64)
  a = pm.arena
  seen = []
  a.on_grad = seen.append
  lp = pm(x, True, use_resnet_d=True)
  pm.backward(torch.zeros((2, 1, 1, pm.ldc), dtype=torch.bfloat16))
  a.on_grad = None
  last = [1 << 62, 1 << 62]
  for off in seen:
    s = 0 if off < a.decay_elems else 1
    assert off <= last[s], 'watermark moved up'
    last[s] = off
  kernels = [sp.offset for n, sp in a.specs.items() if n.endswith('/kernel')]
  assert set(kernels) <= set(seen), 'every kernel gradient is announced'


class _FakeStream(object):
  """stands in for a HIP stream on the CPU double: records who waited for whom"""
  def __init__(self, name, log):
    self.name, self.log = name, log

  def wait_stream(self, other):
    self.log.append(('wait', self.name, other.name))


@pytest.mark.parametrize('name,joins', [('a-r50-d', 6), ('a-r50-beta1-d', 6)])
def test_biglittle_backward_on_two_streams_is_the_same_backward(cpu_double, monkeypatch, name, joins):
  """The big branch's blocks 2..n run their backward interleaved with the little branch's (on the GPU: on the branch stream);
  on the CPU double, with stand-in streams, the reordered tape must give bit-identical gradients, fork / join the streams
  around the interleaved part and keep the gradient-ready watermark monotone with every kernel announced."""
  import contextlib
  from assembled_cnn_amd import model as pmodel

  def run(two_streams):
    log = []
    main, side = _FakeStream('main', log), _FakeStream('side', log)
    cur = [main]

    @contextlib.contextmanager
    def ctx(s):
      log.append(('enter', s.name))
      cur.append(s)
      try:
        yield
      finally:
        cur.pop()
    if two_streams:
      monkeypatch.setattr(pmodel, '_current_stream', lambda: cur[-1])
      monkeypatch.setattr(pmodel, '_stream_ctx', ctx)
      monkeypatch.setattr(pmodel.Model, '_branch_stream', lambda self, c, x: None if c.dry else side)
    _, pm = MP.make_pair(name, 'cpu', 2, 64)
    _, x, _ = MP.inputs(2, 64)
    a = pm.arena
    seen = []
    a.on_grad = seen.append
    lp = pm(x, True, use_resnet_d=True)
    dl = torch.zeros((2, 1, 1, pm.ldc), dtype=torch.bfloat16)
    dl[:, 0, 0, :1001] = (torch.softmax(lp.float(), 1) / 2).to(torch.bfloat16)
    pm.backward(dl)
    a.on_grad = None
    monkeypatch.undo()
    return a, seen, log, a.g32.clone()

  a1, seen1, log1, g1 = run(False)
  a2, seen2, log2, g2 = run(True)
  assert not log1 and torch.equal(g1, g2), 'the two-stream backward must be the same backward'
  last = [1 << 62, 1 << 62]
  for off in seen2:
    s = 0 if off < a2.decay_elems else 1
    assert off <= last[s], 'watermark moved up'
    last[s] = off
  assert sorted(seen1) == sorted(seen2)
  # three BigLittle stages: forward fork + join each, backward fork + several turns on the side stream + join each
  waits = [e for e in log2 if e[0] == 'wait']
  assert waits.count(('wait', 'side', 'main')) == joins and waits.count(('wait', 'main', 'side')) == joins
  assert sum(1 for e in log2 if e == ('enter', 'side')) >= 3 + 3


def test_capture_refuses_what_it_cannot_record(cpu_double):
  """Trainer.capture argument checks that need no GPU: host tensors, an unknown replay mode, DropBlock."""
  from assembled_cnn_amd.train import HParams, Trainer
  hp = dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
            batch_size=2)
  tr = Trainer(HParams(**hp), seed=0, device='cpu')
  img, _, labels = MP.inputs(2, 64)
  assert tr.stream is None
  with pytest.raises(RuntimeError):
    tr.capture(img, labels)                     # device tensors only: there is no CPU path to record
  with pytest.raises(ValueError):
    tr.capture(img, labels, replay='magic')
  db = Trainer(HParams(**dict(hp, use_dropblock=True)), seed=0, device='cpu', recorded=False)
  with pytest.raises(NotImplementedError):
    db.capture(img, labels)                     # no static DropBlock buffers: nothing a replay could rewrite
  tr.release_graph()                            # nothing captured: a no-op
  tr.train_step(img, labels)                    # and the trainer is an ordinary eager trainer


def test_dropblock_from_static_buffers_is_the_eager_dropblock_bit_for_bit(cpu_double):
  """nn.DropBlockState (what makes the published recipe recordable): draws in static buffers rewritten before every step,
  gamma read from device memory, every DropBlock call of the topology issued whatever keep_prob is.  Against the eager
  form (draws handed to the layers, gamma by value): the same losses and the same weights bit for bit over steps whose
  keep_prob moves.  At keep_prob == 1 exactly (the first step of the reference's schedule) the eager form skips DropBlock
  and runs the FUSED block tail, the static form runs the un-fused tail with an all-ones mask: the same function, two more
  bf16 roundings per block -- compared within rounding noise."""
  from assembled_cnn_amd import nn
  from assembled_cnn_amd.train import HParams, Trainer
  hp = dict(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3, use_resnet_d=True,
            batch_size=2, use_dropblock=True, dropblock_kp=[0.9, 0.6], train_epochs=1, num_images_train=8,
            base_learning_rate=0.01, learning_rate_decay_type='fixed')
  img, _, labels = MP.inputs(2, 224)
  eager = Trainer(HParams(**hp), seed=0, device='cpu', recorded=False)
  static = Trainer(HParams(**hp), seed=0, device='cpu', recorded=True)
  assert eager._db is None and isinstance(static._db, nn.DropBlockState)
  # discover the draw shapes with a throw-away trainer (the walk asks for them in creation order)
  probe = Trainer(HParams(**hp), seed=0, device='cpu', recorded=True)
  probe.train_step(img, labels)
  shapes = [tuple(u.shape) for (u, _, _, _) in probe._db.slots]
  assert len(shapes) >= 8
  g = torch.Generator().manual_seed(5)
  kps = []
  for step in range(3):
    kp = eager.keep_prob_fn(eager.global_step)
    kps.append(kp)
    draws = [torch.rand(s, generator=g) for s in shapes]
    le = eager.train_step(img, labels, dropblock_uniforms=draws).clone()
    ls = static.train_step(img, labels, dropblock_uniforms=draws).clone()
    assert torch.equal(le, ls), 'step %d: loss rows differ' % step
    assert torch.equal(eager.model.arena.w32, static.model.arena.w32), 'step %d: weights differ' % step
    assert static.last['keep_prob'] == kp
  assert kps[0] == 0.9 and kps[0] > kps[1] > kps[2]
  assert static._db.known and len(static._db.slots) == len(shapes)
  with pytest.raises(ValueError):
    static.train_step(img, labels, dropblock_uniforms=draws[:-1])     # a draw is missing
  # keep_prob == 1: all-ones masks, scale exactly 1
  hp1 = dict(hp, dropblock_kp=[1.0, 1.0])
  e1 = Trainer(HParams(**hp1), seed=0, device='cpu', recorded=False)
  s1 = Trainer(HParams(**hp1), seed=0, device='cpu', recorded=True)
  l_e, l_s = e1.train_step(img, labels).clone(), s1.train_step(img, labels).clone()
  assert float((l_e - l_s).abs().max()) <= 2e-2 * float(l_e.abs().max())
  assert float(s1._db.gamma[:len(shapes)].abs().max()) == 0.0


def test_kd_input_side_is_library_calls_only(cpu_double, monkeypatch):
  """ADVICE (round 4, high): a recorded step replays what the LIBRARY launched; a framework slice / concatenation inside
  the recorded region would be silently missing from every replay.  Behind split_labels the input side must not call a
  torch kernel: with torch.cat / Tensor.contiguous poisoned, _prepare still runs -- KD with both mixup types -- and
  gives what the reference formula gives (utils/data_util.py:97-158 incl. the :154 quirk)."""
  from assembled_cnn_amd.train import HParams, Trainer
  C, B = 1001, 8
  g = torch.Generator().manual_seed(3)
  img = torch.randint(0, 256, (B, 64, 64, 3), generator=g, dtype=torch.uint8)
  hard = torch.nn.functional.one_hot(torch.randint(1, C, (B,), generator=g), C).float()
  tl = torch.randn((B, C), generator=g) * 2.0
  labels = torch.cat([hard, tl], 1)
  lam1, lam2 = torch.rand(B // 2, generator=g), torch.rand(B // 2, generator=g)
  for mt in (1, 2):
    tr = Trainer(HParams(resnet_version=1, batch_size=B, kd_temp=2.0, mixup_type=mt), seed=0, device='cpu')
    h, t = tr.split_labels(labels)
    with monkeypatch.context() as m:
      import sys

      def guard(orig):
        def f(*a, **k):     # the test double of the library computes with torch; the HOST package must not
          if sys._getframe(1).f_globals.get('__name__', '').startswith('assembled_cnn_amd'):
            raise AssertionError('a framework kernel inside the recordable input side')
          return orig(*a, **k)
        return f
      m.setattr(torch, 'cat', guard(torch.cat))
      m.setattr(torch.Tensor, 'contiguous', guard(torch.Tensor.contiguous))
      m.setattr(torch.Tensor, 'clone', guard(torch.Tensor.clone))
      m.setattr(torch.Tensor, 'to', guard(torch.Tensor.to))
      x, onehot, teacher = tr._prepare(img, h, t, lam1, lam2 if mt == 2 else None)
    p = torch.softmax(tl / 2.0, 1)
    half = B // 2
    l1 = lam1[:, None]
    want1 = l1 * p[:half] + (1 - l1) * p[half:]
    if mt == 1:
      assert teacher.shape == (half, C) and torch.allclose(teacher, want1, atol=1e-6)
    else:
      l2 = lam2[:, None]
      want2 = l2 * hard[:half] + (1 - l2) * torch.flip(p[half:], [0])     # y1, not y1_t: the reference's own quirk
      assert teacher.shape == (B, C)
      assert torch.allclose(teacher[:half], want1, atol=1e-6) and torch.allclose(teacher[half:], want2, atol=1e-6)
    # and against the oracle's mixup (pinned to utils/data_util.mixup run from the reference: tests/test_reference_taps.py)
    from oracle import assembled_oracle as O
    xo = O.mean_image_subtraction(img.float())
    _, oy, ot = O.mixup(xo, hard, lam1, keep_batch_size=(mt == 2), y_t=p, lam2=lam2 if mt == 2 else None)
    assert torch.allclose(onehot, oy, atol=1e-6) and torch.allclose(teacher, ot, atol=1e-6)


def test_self_recording_bookkeeping_without_a_gpu(cpu_double, monkeypatch):
  """Trainer.train_step's own recording (train.Trainer._auto_step) with capture() replaced by a counter: it fires after
  AUTO_WARMUP eager steps of ONE input signature, starts counting again when the shapes change, is never attempted with a
  gradient exchange that cannot replay bucket launches, and a failing capture leaves an eager trainer with ONE warning
  (recorded=True: the exception)."""
  import warnings
  from assembled_cnn_amd.train import HParams, Trainer
  hp = HParams(resnet_version=1, batch_size=2, learning_rate_decay_type='fixed', base_learning_rate=0.01)
  img2, _, lab2 = MP.inputs(2, 64)
  img4, _, lab4 = MP.inputs(4, 64)

  class FakeCuda(object):          # _auto_step only asks the tensors whether they live on the device
    def __init__(self, t):
      self.t = t
    is_cuda = True

    def __getattr__(self, k):
      return getattr(self.t, k)

  # the step itself runs on the CPU double; only the bookkeeping is driven with device-flagged tensors
  tr = Trainer(hp, seed=0, device='cpu')
  calls = []
  monkeypatch.setattr(tr, 'capture', lambda *a, **k: calls.append(tuple(a[0].shape)) or tr)
  tr._auto = True
  n = Trainer.AUTO_WARMUP
  for i in range(n):
    tr._auto_step(FakeCuda(img2), lab2, None, None)
  assert calls == [tuple(img2.shape)] and tr._auto_made
  tr._auto_made = False
  for i in range(n - 1):
    tr._auto_step(FakeCuda(img4), lab4, None, None)
  tr._auto_step(FakeCuda(img2), lab2, None, None)          # the signature changed again: the count starts over
  assert len(calls) == 1
  for i in range(n - 1):
    tr._auto_step(FakeCuda(img2), lab2, None, None)
  assert len(calls) == 2
  tr._auto_step(img2, lab2, None, None)                    # host tensors: nothing to record
  assert len(calls) == 2
  tr.grad_sync = object()                                  # an exchange without launch_recorded: not attempted
  tr._auto_n = n
  tr._auto_step(FakeCuda(img2), lab2, None, None)
  assert len(calls) == 2
  tr.grad_sync = None

  def boom(*a, **k):
    raise RuntimeError('no capture on this box')
  monkeypatch.setattr(tr, 'capture', boom)
  tr._auto_n = n
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    tr._auto_step(FakeCuda(img2), lab2, None, None)
  assert len(w) == 1 and 'staying with the eager step' in str(w[0].message)
  assert tr._auto is False and tr.step_mode.startswith('eager (recording failed')
  tr.train_step(img2, lab2)                                # still an ordinary eager trainer
  strict = Trainer(hp, seed=0, device='cpu', recorded=True)
  monkeypatch.setattr(strict, 'capture', boom)
  strict._auto = True
  for i in range(n - 1):
    strict._auto_step(FakeCuda(img2), lab2, None, None)
  with pytest.raises(RuntimeError):
    strict._auto_step(FakeCuda(img2), lab2, None, None)
