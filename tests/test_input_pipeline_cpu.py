"""Input-pipeline tail: oracle known answers, host window arithmetic, and the host mirror through
the CPU test double (the HIP kernel itself is checked bit-exact in tests/test_gpu_input_pipeline.py)."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as IO


def _img(h, w, seed):
  return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def test_legacy_resize_known_answers():
  # 2x2 -> 4x4, scale 0.5: sources 0, 0.5, 1, 1.5 ; the last sample clamps its upper neighbour (no half pixel)
  a = np.array([[[0.], [10.]], [[20.], [30.]]], dtype=np.float32)
  r = IO.resize_bilinear_legacy(a, 4, 4)[..., 0]
  want = np.array([[0, 5, 10, 10], [10, 15, 20, 20], [20, 25, 30, 30], [20, 25, 30, 30]], dtype=np.float32)
  assert np.array_equal(r, want)
  # 4 -> 2 down-sampling picks sources 0 and 2 exactly (no averaging: the aliasing the reference is known for)
  b = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
  assert np.array_equal(IO.resize_bilinear_legacy(b, 2, 2)[..., 0], np.array([[0, 2], [8, 10]], dtype=np.float32))
  # identity when sizes match; 3 -> 2: sources 0, 1.5
  assert np.array_equal(IO.resize_bilinear_legacy(b, 4, 4), b)
  c = np.array([0., 3., 9.], dtype=np.float32).reshape(1, 3, 1)
  assert np.array_equal(IO.resize_bilinear_legacy(c, 1, 2)[0, :, 0], np.array([0., 6.], dtype=np.float32))


def test_window_arithmetic_matches_oracle_and_reference_rules():
  from assembled_cnn_amd import input_pipeline as P
  assert P.smallest_size_at_least(500, 375, 256) == (341, 256) == IO.smallest_size_at_least(500, 375, 256)
  assert P.smallest_size_at_least(333, 500, 256) == IO.smallest_size_at_least(333, 500, 256)
  w = P.eval_window(500, 375, 224, 224, 0)                 # resize_min = int(224 / 0.875) = 256
  assert (w['resize_h'], w['resize_w'], w['out_y'], w['out_x']) == (341, 256, 58, 16)
  w = P.eval_window(375, 500, 256, 256, 0)                 # imagenet_224_256 at evaluation: int(256/0.875) = 292
  assert (w['resize_h'], w['resize_w']) == IO.smallest_size_at_least(375, 500, 292)
  w = P.eval_window(375, 500, 256, 256, 1)                 # crop_type 1: min side + 1
  assert (w['resize_h'], w['resize_w']) == IO.smallest_size_at_least(375, 500, 257)
  assert P.output_size_and_crop_type('imagenet', True) == (224, 0)
  assert P.output_size_and_crop_type('imagenet_224_256', False) == (256, 0)
  assert P.output_size_and_crop_type('imagenet_224_256', True) == (224, 0)
  assert P.output_size_and_crop_type('imagenet_224_256a', False) == (256, 1)
  assert P.output_size_and_crop_type('imagenet_320', False) == (320, 0)
  assert P.output_size_and_crop_type('imagenet_320a', True) == (320, 1)
  with pytest.raises(NotImplementedError):
    P.output_size_and_crop_type('reid', True)


def test_sampled_boxes_obey_the_reference_ranges():
  from assembled_cnn_amd import input_pipeline as P
  rng = np.random.default_rng(0)
  flips = 0
  for k in range(300):
    H, W = int(rng.integers(40, 600)), int(rng.integers(40, 600))
    win = P.train_window(H, W, 224, 224, rng)
    y, x, h, w = win['crop_y'], win['crop_x'], win['crop_h'], win['crop_w']
    assert 0 <= y and 0 <= x and y + h <= H and x + w <= W and h > 0 and w > 0
    assert h * w >= 0.1 * H * W - 1e-6                     # min_object_covered of the whole-image box
    if (h, w) != (H, W):
      assert 0.05 * H * W <= h * w <= H * W
      assert 0.75 - 0.05 <= w / h <= 1.33 + 0.05             # rounding of width = round(h * ratio)
    flips += win['flip']
  assert 100 < flips < 200
  # use_random_crop=False: min_object_covered = 1.0 -> only the whole image qualifies
  win = P.train_window(300, 400, 224, 224, np.random.default_rng(1), use_random_crop=False)
  assert (win['crop_y'], win['crop_x'], win['crop_h'], win['crop_w']) == (0, 0, 300, 400)


def test_ragged_batch_through_the_host_mirror(cpu_double):
  from assembled_cnn_amd import input_pipeline as P
  sizes = [(375, 500), (500, 333), (256, 256), (231, 640), (1, 1)]
  imgs = [_img(h, w, 10 + k) for k, (h, w) in enumerate(sizes)]
  # evaluation: whole image -> 256-min resize -> central 224 crop (the 1x1 image up-samples a constant)
  out = P.preprocess_batch(imgs, False, 'cpu', preprocessing_type='imagenet')
  assert out.shape == (5, 224, 224, 3) and out.dtype == torch.float32
  for k, im in enumerate(imgs):
    assert np.array_equal(out[k].numpy(), IO.preprocess_eval(im, 224, 224)), k
  out = P.preprocess_batch(imgs[:4], False, 'cpu', preprocessing_type='imagenet_224_256a', subtract_mean=False)
  for k, im in enumerate(imgs[:4]):
    assert np.array_equal(out[k].numpy(), IO.preprocess_eval(im, 256, 256, crop_type=1, subtract_mean=False)), k
  # training: shared windows (sampled box + flip) -> 224x224
  rng = np.random.default_rng(3)
  wins = [P.train_window(im.shape[0], im.shape[1], 224, 224, rng) for im in imgs]
  out = P.preprocess_batch(imgs, True, 'cpu', windows=wins)
  for k, (im, w) in enumerate(zip(imgs, wins)):
    ref = IO.preprocess_train_window(im, (w['crop_y'], w['crop_x'], w['crop_h'], w['crop_w'], w['flip']), 224, 224)
    assert np.array_equal(out[k].numpy(), ref), k
  # empty batch, bad inputs
  assert P.preprocess_batch([], False, 'cpu').shape == (0, 224, 224, 3)
  with pytest.raises(ValueError):
    P.preprocess_batch([np.zeros((4, 4), np.uint8)], False, 'cpu')
  with pytest.raises(ValueError):
    P.preprocess_batch([imgs[0]], True, 'cpu', windows=[dict(crop_y=0, crop_x=0, crop_h=999, crop_w=10, resize_h=224,
                                                             resize_w=224, out_y=0, out_x=0, flip=0)])


def test_pipeline_output_feeds_the_trainer(cpu_double):
  """subtract_mean=False output is what Trainer.train_step takes; subtract_mean=True is what Model() takes."""
  from assembled_cnn_amd import input_pipeline as P, train
  imgs = [_img(90, 120, k) for k in range(4)]
  rng = np.random.default_rng(0)
  wins = [P.train_window(90, 120, 64, 64, rng) for _ in imgs]
  x = P.preprocess_batch(imgs, True, 'cpu', image_size=64, windows=wins, subtract_mean=False)
  hp = train.HParams(zero_gamma=True, learning_rate_decay_type='fixed', base_learning_rate=0.01, batch_size=4)
  tr = train.Trainer(hp, device='cpu')
  loss = tr.train_step(x, torch.tensor([1, 2, 3, 4], dtype=torch.int32))
  assert bool(torch.isfinite(loss).all())
  xm = P.preprocess_batch(imgs, True, 'cpu', image_size=64, windows=wins, subtract_mean=True)
  logits = tr.model(xm, False)
  assert logits.shape == (4, 1001)
