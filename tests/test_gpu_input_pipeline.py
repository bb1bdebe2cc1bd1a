"""HIP input-pipeline tail vs the numpy oracle: bit-exact float32 (legacy bilinear resize has no tolerance)."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as IO

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
  return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def test_eval_ragged_batch_bit_exact(hip_lib):
  from assembled_cnn_amd import input_pipeline as P
  sizes = [(375, 500), (500, 333), (256, 256), (231, 640), (1, 1), (2000, 1500), (224, 224), (97, 1201)]
  imgs = [_img(h, w, 20 + k) for k, (h, w) in enumerate(sizes)]
  for ptype, side, ct in [('imagenet', 224, 0), ('imagenet_224_256', 256, 0), ('imagenet_224_256a', 256, 1),
                          ('imagenet_320', 320, 0)]:
    out = P.preprocess_batch(imgs, False, 'cuda', preprocessing_type=ptype).cpu().numpy()
    assert out.shape == (len(imgs), side, side, 3)
    for k, im in enumerate(imgs):
      assert np.array_equal(out[k], IO.preprocess_eval(im, side, side, crop_type=ct)), (ptype, k)


def test_train_windows_and_flip_bit_exact(hip_lib):
  from assembled_cnn_amd import input_pipeline as P
  rng = np.random.default_rng(7)
  imgs = [_img(int(rng.integers(30, 700)), int(rng.integers(30, 700)), 100 + k) for k in range(24)]
  wins = [P.train_window(im.shape[0], im.shape[1], 224, 224, rng) for im in imgs]
  assert any(w['flip'] for w in wins) and not all(w['flip'] for w in wins)
  for sub in (True, False):
    out = P.preprocess_batch(imgs, True, 'cuda', windows=wins, subtract_mean=sub).cpu().numpy()
    for k, (im, w) in enumerate(zip(imgs, wins)):
      ref = IO.preprocess_train_window(im, (w['crop_y'], w['crop_x'], w['crop_h'], w['crop_w'], w['flip']), 224, 224,
                                       subtract_mean=sub)
      assert np.array_equal(out[k], ref), k


def test_empty_batch_bad_descriptor_and_full_batch_properties(hip_lib):
  from assembled_cnn_amd import input_pipeline as P, ops
  assert P.preprocess_batch([], False, 'cuda').shape == (0, 224, 224, 3)
  # a descriptor that does not fit yields zeros for that image only (the host mirror normally raises first)
  im = _img(50, 60, 1)
  good = P.eval_window(50, 60, 32, 32)
  buf, table = P.pack_batch([im, im], [good, good], 32, 32)
  t = table.clone()
  t.view(torch.int32)[14 + 6] = 9999            # second descriptor: crop_h beyond the image
  out = ops.resize_crop_flip(buf.cuda(), t.cuda(), 2, 32, 32, True).cpu()
  assert np.array_equal(out[0].numpy(), IO.preprocess_eval(im, 32, 32)) and float(out[1].abs().max()) == 0.0
  # BASELINE-size batch (256 images): constant images stay constant, flipping twice is the identity,
  # resizing to the source size is the identity
  consts = [np.full((300 + k, 280, 3), k % 256, np.uint8) for k in range(256)]
  out = P.preprocess_batch(consts, False, 'cuda', subtract_mean=False)
  want = (torch.arange(256) % 256).float().cuda()[:, None, None, None].expand_as(out)
  assert torch.equal(out, want)
  ident = dict(crop_y=0, crop_x=0, crop_h=50, crop_w=60, resize_h=50, resize_w=60, out_y=0, out_x=0, flip=0)
  buf, table = P.pack_batch([im], [dict(ident)], 50, 60)
  same = ops.resize_crop_flip(buf.cuda(), table.cuda(), 1, 50, 60, False).cpu().numpy()[0]
  assert np.array_equal(same, im.astype(np.float32))
  buf, table = P.pack_batch([im], [dict(ident, flip=1)], 50, 60)
  flipped = ops.resize_crop_flip(buf.cuda(), table.cuda(), 1, 50, 60, False).cpu().numpy()[0]
  assert np.array_equal(flipped, im[:, ::-1].astype(np.float32))


def test_hip_pipeline_equals_reference_preprocess_image(hip_lib):
  """asm_resize_crop_flip (window -> flip -> legacy bilinear resize -> central crop -> mean subtraction) against
  preprocessing/imagenet_preprocessing.preprocess_image run from the reference's source under the shim, evaluation
  (two crop types) and training branches, ragged sizes incl. 1 x 1 (tests/golden/reference_step.json)"""
  from tests.test_reference_step import check_preprocessing_against_reference
  check_preprocessing_against_reference('cuda', 'product')
