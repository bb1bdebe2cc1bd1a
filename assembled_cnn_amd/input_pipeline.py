"""GPU tail of the input pipeline.

Mirrors ``utils/data_util.py:267-386 preprocess_image`` + ``preprocessing/imagenet_preprocessing.py:269-313`` from
the point where the reference holds a decoded uint8 image: window selection (random box / whole image), flip,
legacy bilinear resize, central crop and mean subtraction.  JPEG decoding and record parsing stay on the host
(out of scope); the window arithmetic below is host integer/float32 math done exactly as the reference's graph
does it, and the pixel work is one HIP launch for the whole ragged batch (ops.resize_crop_flip).

The output of ``preprocess_batch(..., subtract_mean=True)`` is what the reference's ``preprocess_image`` returns
before ``tf.cast(image, dtype)`` (float32, mean-subtracted, NHWC) and feeds ``Model.__call__`` directly; with
``subtract_mean=False`` it feeds ``Trainer.train_step`` (whose fused mixup kernel subtracts the means).
"""
from __future__ import annotations

import ctypes
import math
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .lib import ImageDesc

_RESIZE_MIN = 256   # preprocessing/imagenet_preprocessing.py:54


def output_size_and_crop_type(preprocessing_type: str, is_training: bool, image_size: int = 224) -> Tuple[int, int]:
  """utils/data_util.py:275-343: (output side, crop_type) for the imagenet* preprocessing types."""
  if preprocessing_type == 'imagenet':
    return image_size, 0
  if preprocessing_type == 'imagenet_224_256':
    return (224 if is_training else 256), 0
  if preprocessing_type == 'imagenet_224_256a':
    return (224 if is_training else 256), 1
  if re.compile('imagenet_[0-9]{3}a').match(preprocessing_type):
    return int(preprocessing_type.split('_')[1][0:3]), 1
  if re.compile('imagenet_[0-9]{3}').match(preprocessing_type):
    return int(preprocessing_type.split('_')[1]), 0
  raise NotImplementedError('preprocessing_type %r (only the imagenet* family is on the hot path)' % preprocessing_type)


def smallest_size_at_least(height: int, width: int, resize_min) -> Tuple[int, int]:
  """preprocessing/imagenet_preprocessing.py:158-186, in float32 like the graph."""
  resize_min = np.float32(resize_min)
  h, w = np.float32(height), np.float32(width)
  scale_ratio = resize_min / np.minimum(h, w)
  return int(np.float32(h * scale_ratio)), int(np.float32(w * scale_ratio))


def eval_window(height: int, width: int, out_h: int, out_w: int, crop_type: int = 0):
  """Whole image -> aspect-preserving resize -> central crop (imagenet_preprocessing.py:303-310, :97-120)."""
  if crop_type == 1:
    resize_min = int(min(out_h, out_w) + 1)
  else:
    resize_min = int(min(out_h, out_w) * (1.0 / 0.875))
  rh, rw = smallest_size_at_least(height, width, resize_min)
  if rh < out_h or rw < out_w:
    raise ValueError('resized image %dx%d is smaller than the %dx%d crop' % (rh, rw, out_h, out_w))
  return dict(crop_y=0, crop_x=0, crop_h=height, crop_w=width, resize_h=rh, resize_w=rw,
              out_y=(rh - out_h) // 2, out_x=(rw - out_w) // 2, flip=0)


def sample_distorted_bounding_box(height: int, width: int, rng: np.random.Generator, min_object_covered=0.1,
                                  aspect_ratio_range=(0.75, 1.33), area_range=(0.05, 1.0), max_attempts=100):
  """tf.image.sample_distorted_bounding_box with no annotated box and use_image_if_no_bounding_boxes=True
  (imagenet_preprocessing.py:66-76).  TensorFlow 1.14's published algorithm (sample_distorted_bounding_box_op.cc,
  GenerateRandomCrop): draw an aspect ratio, derive the admissible height range from the area range, draw a height,
  width = round(height * ratio), fix up the area by one row, draw the offsets; accept the first attempt that
  covers at least min_object_covered of the (whole-image) box, else fall back to the whole image.  TF's own
  random stream cannot be reproduced; `rng` supplies the draws.  Returns (y, x, h, w)."""
  lo, hi = aspect_ratio_range
  min_area = area_range[0] * width * height
  max_area = area_range[1] * width * height
  eps = 1e-7
  for _ in range(max_attempts):
    ratio = float(rng.uniform(lo, hi))
    h = int(round(math.sqrt(min_area / ratio)))
    max_h = int(round(math.sqrt(max_area / ratio)))
    if int(round(max_h * ratio)) > width:
      max_h = int((width + 0.5 - eps) / ratio)
    max_h = min(max_h, height)
    h = min(h, max_h)
    if h < max_h:
      h += int(rng.integers(0, max_h - h + 1))
    w = int(round(h * ratio))
    area = w * h
    if area < min_area:
      h += 1
      w = int(round(h * ratio))
      area = w * h
    if area > max_area:
      h -= 1
      w = int(round(h * ratio))
      area = w * h
    if area < min_area or area > max_area or w > width or h > height or w <= 0 or h <= 0:
      continue
    y = int(rng.integers(0, height - h + 1)) if h < height else 0
    x = int(rng.integers(0, width - w + 1)) if w < width else 0
    if area >= min_object_covered * width * height:     # coverage of the whole-image box by the crop
      return y, x, h, w
  return 0, 0, height, width


def train_window(height: int, width: int, out_h: int, out_w: int, rng: np.random.Generator,
                 use_random_crop: bool = True):
  """_decode_crop_and_flip + _resize_image (imagenet_preprocessing.py:57-97, :283-287)."""
  y, x, h, w = sample_distorted_bounding_box(height, width, rng,
                                             min_object_covered=0.1 if use_random_crop else 1.0)
  flip = int(rng.random() < 0.5)     # tf.image.random_flip_left_right
  return dict(crop_y=y, crop_x=x, crop_h=h, crop_w=w, resize_h=out_h, resize_w=out_w, out_y=0, out_x=0, flip=flip)


def _validate(win: dict, Hs: int, Ws: int, out_h: int, out_w: int):
  ok = (win['crop_h'] > 0 and win['crop_w'] > 0 and win['crop_y'] >= 0 and win['crop_x'] >= 0 and
        win['crop_y'] + win['crop_h'] <= Hs and win['crop_x'] + win['crop_w'] <= Ws and win['resize_h'] > 0 and
        win['resize_w'] > 0 and win['out_y'] >= 0 and win['out_x'] >= 0 and
        win['out_y'] + out_h <= win['resize_h'] and win['out_x'] + out_w <= win['resize_w'])
  if not ok:
    raise ValueError('window %r does not fit a %dx%d image / %dx%d output' % (win, Hs, Ws, out_h, out_w))


_DESC_DTYPE = np.dtype([('src_offset', '<i8'), ('Hs', '<i4'), ('Ws', '<i4'), ('crop_y', '<i4'), ('crop_x', '<i4'),
                        ('crop_h', '<i4'), ('crop_w', '<i4'), ('resize_h', '<i4'), ('resize_w', '<i4'),
                        ('out_y', '<i4'), ('out_x', '<i4'), ('flip', '<i4'), ('reserved', '<i4')])
assert _DESC_DTYPE.itemsize == ctypes.sizeof(ImageDesc) == 56


_STAGING = {}


def _staging(nbytes: int, pin: bool) -> torch.Tensor:
  """Grow-only host staging buffer (re-used across batches: a fresh 150 MB allocation costs more in page faults than
  the copy itself).  The caller must have consumed the previous batch's H2D copy before packing the next one."""
  ev = _STAGING.get('event')
  if pin and ev is not None:
    ev.synchronize()            # the previous batch's asynchronous H2D copy reads this buffer
  cur = _STAGING.get(pin)
  if cur is None or cur.numel() < nbytes:
    cur = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=pin)
    _STAGING[pin] = cur
  return cur


def pack_batch(images: Sequence[np.ndarray], windows: Sequence[dict], out_h: int, out_w: int, pin: bool = False):
  """Decoded images (uint8 [H, W, 3], any sizes) -> (packed uint8 buffer, descriptor table bytes) as CPU tensors
  (pinned when ``pin``).  One memcpy per image into the staging buffer and a vectorised descriptor table: ~10 GB/s
  on one core, so the host side keeps up with the GPU (the per-image tensor ops of the first version did 670 img/s)."""
  n = len(images)
  if n != len(windows):
    raise ValueError('one window per image')
  sizes = np.empty(n, dtype=np.int64)
  for k, im in enumerate(images):
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
      raise ValueError('Input must be of size [height, width, 3] uint8')     # imagenet_preprocessing.py:143-144
    sizes[k] = im.size
  padded = (sizes + 15) // 16 * 16
  offs = np.concatenate([[0], np.cumsum(padded)[:-1]]) if n else np.zeros(0, np.int64)
  total = int(padded.sum()) if n else 0
  buf = _staging(max(total, 16), pin)
  dst = buf.numpy()
  table = torch.zeros(max(n, 1) * 56, dtype=torch.uint8, pin_memory=pin)
  desc = table.numpy().view(_DESC_DTYPE)
  for k, (im, win) in enumerate(zip(images, windows)):
    _validate(win, im.shape[0], im.shape[1], out_h, out_w)
    o = int(offs[k])
    dst[o:o + im.size] = im.reshape(-1)                                       # one memcpy (copies if not contiguous)
  if n:
    desc['src_offset'][:n] = offs
    desc['Hs'][:n] = [im.shape[0] for im in images]
    desc['Ws'][:n] = [im.shape[1] for im in images]
    for f in ('crop_y', 'crop_x', 'crop_h', 'crop_w', 'resize_h', 'resize_w', 'out_y', 'out_x', 'flip'):
      desc[f][:n] = [int(w[f]) for w in windows]
  buf = buf[:max(total, 16)]
  return buf, table[:56 * n]


def preprocess_batch(images: Sequence[np.ndarray], is_training: bool, device, image_size: int = 224,
                     preprocessing_type: str = 'imagenet', use_random_crop: bool = True,
                     rng: Optional[np.random.Generator] = None, subtract_mean: bool = True,
                     windows: Optional[List[dict]] = None) -> torch.Tensor:
  """The imagenet* branches of data_util.preprocess_image for a batch of decoded images; float32 NHWC on `device`.
  `windows` overrides the sampled / computed windows (tests share them with the oracle)."""
  side, crop_type = output_size_and_crop_type(preprocessing_type, is_training, image_size)
  if windows is None:
    if is_training:
      rng = rng if rng is not None else np.random.default_rng()
      windows = [train_window(im.shape[0], im.shape[1], side, side, rng, use_random_crop) for im in images]
    else:
      windows = [eval_window(im.shape[0], im.shape[1], side, side, crop_type) for im in images]
  dev = torch.device(device)
  buf, table = pack_batch(images, windows, side, side, pin=dev.type == 'cuda')
  bd, td = buf.to(dev, non_blocking=True), table.to(dev, non_blocking=True)
  if dev.type == 'cuda':
    ev = torch.cuda.Event()
    ev.record()
    _STAGING['event'] = ev
  return ops.resize_crop_flip(bd, td, len(images), side, side, subtract_mean)
