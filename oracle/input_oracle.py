"""ORACLE (test infrastructure only -- never imported by the product path): numpy restatement of the tensor part of
the reference's ImageNet preprocessing.  PINNED (round 3) to the reference's own source: preprocess_image (evaluation
with both crop types, training with recorded box / flip draws), central_crop, _smallest_size_at_least and
mean_image_subtraction of preprocessing/imagenet_preprocessing.py are executed UNMODIFIED under oracle/tf_shim in
float32 (tests/golden/make_reference_step.py -> tests/golden/reference_step.json) and tests/test_reference_step.py
requires this module to reproduce them.  What stays [TF-sem] is the inside of tf.image.resize_images: TF-1.14's legacy
bilinear kernel (resize_bilinear_op.cc: scale = in / out, no half-pixel centres, compute_lerp order), restated here
and -- independently, in general form -- in the shim, plus hand-computed examples in tests/test_input_pipeline_cpu.py.

  resize_bilinear_legacy   tf.image.resize_images(BILINEAR, align_corners=False)   imagenet_preprocessing.py:210-225
  smallest_size_at_least   imagenet_preprocessing.py:158-186
  central_crop             imagenet_preprocessing.py:97-120
  mean_image_subtraction   imagenet_preprocessing.py:122-155
  preprocess_eval / preprocess_train_window    imagenet_preprocessing.py:269-313
"""
import numpy as np

CHANNEL_MEANS = np.array([123.68, 116.78, 103.94], dtype=np.float32)   # imagenet_preprocessing.py:46-49


def resize_bilinear_legacy(image, out_h, out_w):
  """image: [H, W, C] (uint8 or float).  Float32 arithmetic in the order of TF's compute_lerp."""
  img = np.asarray(image).astype(np.float32)
  H, W = img.shape[:2]

  def weights(in_size, out_size):
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.arange(out_size, dtype=np.float32) * scale
    lower = src.astype(np.int64)
    upper = np.minimum(lower + 1, in_size - 1)
    return lower, upper, (src - lower.astype(np.float32)).astype(np.float32)

  ly, uy, fy = weights(H, out_h)
  lx, ux, fx = weights(W, out_w)
  fx = fx[None, :, None]
  fy = fy[:, None, None]
  tl, tr = img[ly][:, lx], img[ly][:, ux]
  bl, br = img[uy][:, lx], img[uy][:, ux]
  top = tl + (tr - tl) * fx
  bot = bl + (br - bl) * fx
  return (top + (bot - top) * fy).astype(np.float32)


def smallest_size_at_least(height, width, resize_min):
  resize_min = np.float32(resize_min)
  h, w = np.float32(height), np.float32(width)
  scale_ratio = resize_min / np.minimum(h, w)
  return int(np.float32(h * scale_ratio)), int(np.float32(w * scale_ratio))


def central_crop(image, crop_h, crop_w):
  H, W = image.shape[:2]
  top, left = (H - crop_h) // 2, (W - crop_w) // 2
  return image[top:top + crop_h, left:left + crop_w]


def mean_image_subtraction(image):
  if image.ndim != 3:
    raise ValueError('Input must be of size [height, width, C>0]')
  return (image - CHANNEL_MEANS).astype(np.float32)


def preprocess_eval(image, out_h, out_w, crop_type=0, subtract_mean=True):
  resize_min = int(min(out_h, out_w) + 1) if crop_type == 1 else int(min(out_h, out_w) * (1.0 / 0.875))
  rh, rw = smallest_size_at_least(image.shape[0], image.shape[1], resize_min)
  out = central_crop(resize_bilinear_legacy(image, rh, rw), out_h, out_w)
  return mean_image_subtraction(out) if subtract_mean else out


def preprocess_train_window(image, window, out_h, out_w, subtract_mean=True):
  """window = (y, x, h, w, flip): decode_and_crop -> flip_left_right -> resize -> mean subtraction."""
  y, x, h, w, flip = window
  crop = image[y:y + h, x:x + w]
  if flip:
    crop = crop[:, ::-1]
  out = resize_bilinear_legacy(crop, out_h, out_w)
  return mean_image_subtraction(out) if subtract_mean else out
