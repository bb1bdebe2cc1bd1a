"""GPU parity of the widened rows: DropBlock, GeM, sigmoid loss, evaluation metrics kernels vs the oracle,
and DropBlock / sigmoid / GeM through the whole model."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import model_parity as mp
from tests import util

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
  return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(BF)


@pytest.mark.parametrize('H,W,Cn,kp,gs', [(14, 14, 64, 0.9, 0.25), (7, 7, 128, 0.8, 1.0), (9, 12, 8, 0.7, 1.0)])
def test_dropblock_kernels(hip_lib, H, W, Cn, kp, gs):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  N, bs = 3, 7
  x = _rand((N, H, W, Cn), 1)
  u = torch.rand((1, Cn, H - bs + 1, W - bs + 1), generator=torch.Generator().manual_seed(2))
  xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  yr = O.dropblock(xr, kp, bs, gs, True, u)
  gamma = gs * (1. - kp) * (W * H) / (bs ** 2) / ((W - bs + 1) * (H - bs + 1))
  keep, scale = ops.dropblock_mask(u[0].permute(1, 2, 0).contiguous().cuda(), float(gamma), H, W, Cn, bs)
  y = ops.dropblock_apply(x.cuda(), keep, scale)
  assert util.rel_l2(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) <= 4e-3
  dy = _rand((N, H, W, Cn), 3)
  (gx,) = torch.autograd.grad(yr, xr, dy.float().permute(0, 3, 1, 2))
  dx = ops.dropblock_apply(dy.cuda(), keep, scale)
  assert util.rel_l2(dx.float().cpu(), gx.permute(0, 2, 3, 1)) <= 4e-3
  # fused ReLU variant
  yrelu = ops.dropblock_apply(x.cuda(), keep, scale, relu=True)
  assert torch.equal(yrelu, ops.relu_fwd(y))
  dxr = ops.dropblock_apply(dy.cuda(), keep, scale, relu_mask_from=yrelu)
  assert torch.equal(dxr, torch.where(yrelu > 0, dx, torch.zeros_like(dx)))


def test_gem_kernels(hip_lib):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  x = _rand((3, 7, 7, 256), 1)
  xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  yr = O.generalized_mean_pooling(xr)
  y, ssum = ops.gem_fwd(x.cuda())
  assert util.rel_l2(y.float().cpu().view(3, 256), yr.detach().view(3, 256)) <= 4e-3
  dy = _rand((3, 1, 1, 256), 2)
  (gx,) = torch.autograd.grad(yr, xr, dy.float().view(3, 256, 1, 1))
  dx = ops.gem_bwd(x.cuda(), dy.cuda(), ssum)
  assert util.rel_l2(dx.float().cpu(), gx.permute(0, 2, 3, 1)) <= 6e-3


def test_sigmoid_ce_kernel(hip_lib):
  from assembled_cnn_amd import ops
  from oracle import assembled_oracle as O
  B, Cn, ld = 17, 1001, 1008
  g = torch.Generator().manual_seed(1)
  logits = torch.zeros(B, ld)
  logits[:, :Cn] = torch.randn(B, Cn, generator=g) * 3
  y = F.one_hot(torch.randint(0, Cn, (B,), generator=g), Cn).float()
  z = logits[:, :Cn].clone().requires_grad_(True)
  loss = O.get_sup_loss(z, y, 'sigmoid')
  (gz,) = torch.autograd.grad(loss * 4.0, z)
  out, dz = ops.sigmoid_ce(logits.cuda(), ld, y.cuda(), B, Cn, 4.0, ld)
  assert abs(float(out[0]) - float(loss)) <= 1e-5 * abs(float(loss)) and float(out[1]) == B
  dz = dz.view(B, ld).float().cpu()
  assert float(dz[:, Cn:].abs().max()) == 0.0
  assert util.rel_l2(dz[:, :Cn], gz) <= 4e-3


def test_eval_metric_kernels(hip_lib):
  from assembled_cnn_amd import ops
  B, Cn, ld = 64, 1001, 1008
  g = torch.Generator().manual_seed(3)
  logits = torch.zeros(B, ld)
  logits[:, :Cn] = torch.randn(B, Cn, generator=g) * 2
  labels = torch.randint(0, Cn, (B,), generator=g).to(torch.int32)
  labels[:20] = logits[:20, :Cn].argmax(1).to(torch.int32)            # some hits
  pred, conf, top1, top5 = ops.eval_rows(logits.cuda(), ld, labels.cuda(), B, Cn)
  z = logits[:, :Cn]
  assert torch.equal(pred.cpu().long(), z.argmax(1))
  assert torch.allclose(conf.cpu(), torch.softmax(z, 1).max(1).values, rtol=1e-4)
  assert torch.equal(top1.cpu(), (z.argmax(1) == labels.long()).float())
  in5 = torch.tensor([labels[b] in z[b].topk(5).indices.tolist() for b in range(B)]).float()
  assert torch.equal(top5.cpu(), in5)
  state = torch.zeros(33, device='cuda')
  ops.eval_accumulate(conf, top1, top5, state)
  ops.eval_accumulate(conf, top1, top5, state)
  st = state.cpu()
  assert float(st[2]) == 2 * B and abs(float(st[0]) - 2 * float(top1.sum())) < 1e-4
  assert abs(float(st[23:33].sum()) - 2 * B) < 1e-4 and abs(float(st[13:23].sum()) - 2 * float(conf.sum())) < 1e-3


def test_dropblock_model_forward(hip_lib):
  from tests.test_next_rows_cpu import _db_provider
  om, pm = mp.make_pair('a-r50', 'cuda', 4, 224)
  _, x, _ = mp.inputs(4, 224)
  log = []
  lo = om(x, True, keep_prob=0.8, dropblock_uniforms=_db_provider(5, log)).detach()
  uni = [u[0].permute(1, 2, 0).contiguous().cuda() for u in log]
  lp = pm(x.cuda(), True, keep_prob=0.8, dropblock_uniforms=uni).float().cpu()
  assert util.rel_l2(lp, lo) <= 8e-2
  # device-generated draws (the production path) run and change the output
  lq = pm(x.cuda(), True, keep_prob=0.8).float().cpu()
  assert bool(torch.isfinite(lq).all()) and not torch.equal(lq, lp)


def test_assemble_recipe_step_with_dropblock(hip_lib):
  """scripts/train_assemble_from_scratch.sh flag set: BL + SK + sconv3 + mixup + label smoothing + KD + DropBlock."""
  from assembled_cnn_amd import train
  hp = train.HParams(resnet_version=2, use_sk_block=True, anti_alias_type='sconv', anti_alias_filter_size=3,
                     zero_gamma=True, use_dropblock=True, dropblock_kp=[0.9, 0.7], mixup_type=1, label_smoothing=0.1,
                     kd_temp=1.0, learning_rate_decay_type='cosine', lr_warmup_epochs=5, train_epochs=600,
                     base_learning_rate=0.4, weight_decay=1e-4, batch_size=4, num_images_train=1281167)
  tr = train.Trainer(hp, device='cuda')
  img, _, labels = mp.inputs(8, 224)
  g = torch.Generator().manual_seed(0)
  lab = torch.cat([F.one_hot(labels.long(), 1001).float(), torch.randn(8, 1001, generator=g) * 3], 1).cuda()
  lam = tr.sample_mixup_lambdas(4, rng=np.random.default_rng(0))
  l0 = float(tr.train_step(img.cuda(), lab, lam).mean())
  for _ in range(3):
    tr.train_step(img.cuda(), lab, lam, lr=0.01)
  l1 = float(tr.last['loss_rows'].mean())
  assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0


def test_ece_kernel_equals_reference_ece_metric(hip_lib):
  """asm_eval_accumulate's per-bin counts / correct counts / confidence sums and the streaming ECE after each of
  three batches == metric/ece_metric.py run from the reference's source (tests/golden/reference_step.json), incl.
  confidences exactly on the bin edges"""
  from tests.test_reference_step import check_ece_against_reference
  check_ece_against_reference('cuda')
