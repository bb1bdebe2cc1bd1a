"""Host logic of assembled_cnn_amd (arenas, tape, topology walker, trainer) driven through the CPU test
double of the C ABI and compared with the oracle.  No GPU; the HIP kernels themselves are covered by the
-m gpu tier."""
import pytest
import torch

from tests import model_parity as mp


@pytest.mark.parametrize('name', ['r50v1', 'a-r50-d', 'se-proj'])
def test_forward_train_mode(cpu_double, name):
  mp.check_forward(name, 'cpu', 8, 64, True, 6e-2)


def test_forward_eval_mode_uses_moving_stats(cpu_double):
  mp.check_forward('a-r50', 'cpu', 4, 64, False, 4e-2)


@pytest.mark.parametrize('name', ['a-r50', 'r50v1-d'])
def test_backward_tape_vs_autograd(cpu_double, name):
  mp.check_backward(name, 'cpu', 8, 64)


def test_forward_backward_literal_zero_gamma(cpu_double):
  """the recipe's zero_gamma=True with the block-final gammas left at 0 (not the 0.25 the other whole-model tests damp them
  to): every residual branch is switched off in the forward pass and receives an exactly-zero gradient"""
  mp.check_forward('a-r50-d', 'cpu', 4, 64, True, 6e-2, damp=None)
  mp.check_backward('a-r50-d', 'cpu', 4, 64, damp=None)


def test_train_steps_mixup_label_smoothing(cpu_double):
  mp.check_train_steps('a-r50', 'cpu', 4, 64, 3, dict(base_learning_rate=0.001, weight_decay=1e-4, label_smoothing=0.1),
                       mixup_type=1, rel_tol=3e-2, state_tol=5e-2, mom_cos=0.4)     # (batch 4 at 64 x 64: 16 - 64 samples per channel in the deep stages)


def test_train_steps_kd(cpu_double):
  mp.check_train_steps('r50v1', 'cpu', 4, 64, 2, dict(base_learning_rate=0.001, weight_decay=1e-4), kd_temp=1.0,
                       rel_tol=3e-2, state_tol=5e-2, mom_cos=0.4)


def test_variable_names_counts_and_flag_errors(cpu_double):
  from assembled_cnn_amd.model import Model
  m = Model(50, num_classes=1001, device='cpu', resnet_version=2, use_sk_block=True, anti_alias_type='sconv',
            anti_alias_filter_size=3)
  m.build((64, 64))
  assert m.num_params() == 41848489 and len(m.arena.specs) == 306
  assert list(m.arena.specs)[0] == 'resnet_model/stage0/conv2d/kernel'
  assert 'resnet_model/stage1/big1/sk_block/sk_fc_1/kernel' in m.arena.specs
  with pytest.raises(ValueError):
    Model(50, num_classes=10, resnet_version=3)
  with pytest.raises(ValueError):
    Model(77, num_classes=10)
  with pytest.raises(NotImplementedError):
    Model(18, num_classes=10)
  with pytest.raises(NotImplementedError):
    Model(50, num_classes=10, dtype='fp32')
  with pytest.raises(ValueError):
    Model(50, num_classes=10, dtype='int8')
  with pytest.raises(NotImplementedError):
    Model(50, num_classes=10, pool_type='nope')
  with pytest.raises(ValueError):
    m(torch.zeros(1, 64, 64, 3), False, use_resnet_d=True)   # variables were created without the D stem


def test_assemble_r152_variables_match_oracle(cpu_double):
  """BASELINE config 5 topology (A-R152, alpha 1, beta 2): same names, order and shapes as the oracle walk."""
  from assembled_cnn_amd.model import Model, hwio_to_krsc
  from oracle import assembled_oracle as O
  kw = mp.CONFIGS['a-r152']
  om = O.Model(num_classes=1001, **kw)
  om(torch.zeros(1, 64, 64, 3), True)
  pm = Model(num_classes=1001, device='cpu', **kw)
  pm.build((64, 64))
  assert pm.num_params() == 117006249 and len(pm.arena.specs) == 969
  assert list(pm.arena.specs) == list(om.vars.trainable)
  for n, t in om.vars.trainable.items():
    want = tuple(hwio_to_krsc(t).shape) if t.dim() == 4 else (tuple(t.shape) if t.dim() == 1 else None)
    if want is not None:
      assert tuple(pm.arena.w(n).shape) == want, n


def test_lr_schedule_and_hparams_match_reference_defaults():
  from assembled_cnn_amd import train
  from oracle import assembled_oracle as O
  p = train.HParams()
  assert (p.resnet_version, p.bn_momentum, p.weight_decay, p.momentum, p.base_learning_rate,
          p.learning_rate_decay_type, p.bl_alpha, p.bl_beta, p.mixup_type, p.kd_temp, p.label_smoothing) == (
              1, 0.997, 4e-5, 0.9, 0.01, 'exponential', 2, 4, 0, 0.0, 0.0)
  assert p.get_loss_scale() == 1.0 and train.HParams(dtype='fp16').get_loss_scale() == 128.0
  assert train.HParams(loss_scale=64).get_loss_scale() == 64.0
  for kind in ('exponential', 'fixed', 'polynomial', 'piecewise', 'cosine'):
    args = (kind, 1024, 1024, 1281167, 2.0, 0.94, 1e-4, [30, 60, 80, 90], [1, .1, .01, .001, 1e-4], 0.4)
    a = train.learning_rate_with_decay(*args, warmup_epochs=5, train_epochs=120)
    b = O.learning_rate_with_decay(*args, warmup_epochs=5, train_epochs=120)
    for step in (0, 1, 100, 6255, 6256, 40000, 75000, 150000, 10 ** 7):
      assert abs(a(step) - b(step)) <= 1e-12, (kind, step)
  with pytest.raises(NotImplementedError):
    train.learning_rate_with_decay('nope', 1, 1, 1, 1, 1, 1, [], [], 1)


def test_teacher_forced_per_layer_parity_on_the_double(cpu_double):
  """The per-layer teacher-forcing harness itself (tests/model_parity.check_teacher_forced) through the host code:
  every fused group of Assemble-ResNet-50 + D is matched by variable name and compared."""
  from tests import model_parity as mp
  errs = mp.check_teacher_forced('a-r50-d', 'cpu', 4, 64)
  assert len(errs) >= 180 and max(e[2] for e in errs) <= 4e-3


def test_teacher_forced_backward_parity_on_the_double(cpu_double):
  """The per-layer BACKWARD harness (tests/model_parity.check_teacher_forced_backward) through the host code of
  Assemble-ResNet-50 + D with every default fusion on: every group's accumulated output gradient is compared with the
  oracle's autograd value and then replaced by it, every variable's gradient is compared at the end."""
  from tests import model_parity as mp
  errs, st = mp.check_teacher_forced_backward('a-r50-d', 'cpu', 4, 64)
  k = st['kinds']
  assert st['forced'] >= 80 and k['dout'] >= 60 and k['dW'] + k['dW-squeeze'] == 117 and len(errs) >= 400


def test_step_graph_capture_needs_a_gpu(cpu_double):
  """Trainer.capture is a device feature: host tensors are refused (no CPU fallback path to capture)"""
  from assembled_cnn_amd.train import HParams, Trainer
  tr = Trainer(HParams(resnet_version=1, batch_size=2), seed=0, device='cpu')
  img, _, labels = mp.inputs(2, 32)
  with pytest.raises(RuntimeError):
    tr.capture(img, labels)
