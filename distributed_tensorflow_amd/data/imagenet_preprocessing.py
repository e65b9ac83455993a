"""ImageNet TFRecord input pipeline with the "ResNet preprocessing".

Parity with the reference `image_classification/imagenet_preprocessing.py`:

* files `train-%05d-of-01024` / `validation-%05d-of-00128` (`:144-153`), sharded by
  input pipeline, file shuffle, 10-way interleave (`:289-310`), 10 000-record
  shuffle buffer (`:52`);
* `tf.train.Example` with `image/encoded`, `image/class/label`,
  `image/object/bbox/{xmin,ymin,xmax,ymax}` (`:156-223`);
* training: sample a distorted bounding box (min_object_covered 0.1, aspect ratio
  [0.75, 1.33], area [0.05, 1], 100 attempts, whole image if no box; `:351-358`),
  decode + crop, random flip, bilinear resize to 224 (`:366-372,483-500`);
* eval: aspect-preserving resize to a 256-pixel short side, central 224 crop
  (`:375-394,433-480,529-537`);
* both: subtract the channel means [123.68, 116.78, 103.94] (`:54-57,397-430`);
  label - 1, float32, shape [1] (`:252-255`).

JPEG decoding and resizing use Pillow (libjpeg-turbo); the reference's TF bilinear
kernel (`align_corners=False`, no antialiasing) differs from Pillow's in the last
bits on down-scaling - pixel-exact parity with TF resizing is unpinned.
"""
from __future__ import annotations

import io
import math
import os
from typing import Iterator, Optional, Tuple

import numpy as np

from ..utils import logging
from ..utils import tfrecord
from . import pipeline

DEFAULT_IMAGE_SIZE = 224
NUM_CHANNELS = 3
NUM_CLASSES = 1001

NUM_IMAGES = {
    "train": 1281167,
    "validation": 50000,
}

_NUM_TRAIN_FILES = 1024
_NUM_VAL_FILES = 128
_SHUFFLE_BUFFER = 10000

_R_MEAN = 123.68
_G_MEAN = 116.78
_B_MEAN = 103.94
_CHANNEL_MEANS = [_R_MEAN, _G_MEAN, _B_MEAN]

# The lower bound for the smallest side of the image for aspect-preserving resizing.
_RESIZE_MIN = 256


def get_filenames(is_training: bool, data_dir: str):
  """Return filenames for dataset."""
  if is_training:
    return [os.path.join(data_dir, "train-%05d-of-01024" % i) for i in range(_NUM_TRAIN_FILES)]
  return [os.path.join(data_dir, "validation-%05d-of-00128" % i) for i in range(_NUM_VAL_FILES)]


def read_tfrecords(path: str) -> Iterator[bytes]:
  """Record payloads of one shard; framing and CRC-32C checks run in the C++ module when it
  is built (a shard is ~140 MB: the pure-Python checksum would dominate the input pipeline)."""
  from . import native_loader  # pylint: disable=import-outside-toplevel
  if native_loader.available():
    yield from native_loader.read_tfrecords(path, verify=True)
    return
  with open(path, "rb") as f:
    yield from tfrecord.read_records(f)


def _parse_example_proto(example_serialized: bytes):
  """Returns (jpeg bytes, int label, bbox array [num_boxes, 4] as ymin,xmin,ymax,xmax)."""
  feats = tfrecord.decode_example(example_serialized)
  enc = feats.get("image/encoded", [b""])
  image_buffer = enc[0] if enc else b""
  lab = feats.get("image/class/label", [-1])
  label = int(lab[0]) if lab else -1
  xmin = feats.get("image/object/bbox/xmin", [])
  ymin = feats.get("image/object/bbox/ymin", [])
  xmax = feats.get("image/object/bbox/xmax", [])
  ymax = feats.get("image/object/bbox/ymax", [])
  n = min(len(xmin), len(ymin), len(xmax), len(ymax))
  bbox = np.array([[ymin[i], xmin[i], ymax[i], xmax[i]] for i in range(n)],
                  dtype=np.float32).reshape(n, 4)
  return image_buffer, label, bbox


def sample_distorted_bounding_box(height: int, width: int, bboxes: np.ndarray, rng,
                                  min_object_covered=0.1, aspect_ratio_range=(0.75, 1.33),
                                  area_range=(0.05, 1.0), max_attempts=100
                                  ) -> Tuple[int, int, int, int]:
  """Returns (offset_y, offset_x, target_height, target_width).

  Same sampling scheme as `tf.image.sample_distorted_bounding_box` with
  `use_image_if_no_bounding_boxes=True`: draw an aspect ratio and an area, place
  the window uniformly, accept when it covers at least `min_object_covered` of
  some annotated box; fall back to the whole image after `max_attempts`.
  """
  if bboxes is None or len(bboxes) == 0:
    bboxes = np.array([[0.0, 0.0, 1.0, 1.0]], dtype=np.float32)
  boxes_px = np.stack([bboxes[:, 0] * height, bboxes[:, 1] * width,
                       bboxes[:, 2] * height, bboxes[:, 3] * width], axis=1)
  img_area = float(height * width)
  for _ in range(max_attempts):
    ratio = rng.uniform(aspect_ratio_range[0], aspect_ratio_range[1])
    min_area = area_range[0] * img_area
    max_area = area_range[1] * img_area
    # height range allowed by the area range at this aspect ratio (width = ratio * height)
    min_h = int(math.ceil(math.sqrt(min_area / ratio)))
    max_h = int(math.floor(math.sqrt(max_area / ratio)))
    max_h = min(max_h, height, int(math.floor(width / ratio)))
    if max_h < min_h or max_h <= 0:
      continue
    h = int(rng.integers(min_h, max_h + 1))
    w = int(round(h * ratio))
    if w <= 0 or w > width or h > height:
      continue
    y = int(rng.integers(0, height - h + 1))
    x = int(rng.integers(0, width - w + 1))
    # coverage of each annotated box by the window
    iy0 = np.maximum(boxes_px[:, 0], y)
    ix0 = np.maximum(boxes_px[:, 1], x)
    iy1 = np.minimum(boxes_px[:, 2], y + h)
    ix1 = np.minimum(boxes_px[:, 3], x + w)
    inter = np.clip(iy1 - iy0, 0, None) * np.clip(ix1 - ix0, 0, None)
    area = np.clip((boxes_px[:, 2] - boxes_px[:, 0]) * (boxes_px[:, 3] - boxes_px[:, 1]),
                   1e-8, None)
    if np.any(inter / area >= min_object_covered):
      return y, x, h, w
  return 0, 0, height, width


def _decode(image_buffer: bytes):
  from PIL import Image  # pylint: disable=import-outside-toplevel
  img = Image.open(io.BytesIO(image_buffer))
  return img if img.mode == "RGB" else img.convert("RGB")     # CMYK / grayscale files exist


def _jpeg_draft_enabled() -> bool:
  """Scaled JPEG decoding of large training crops: an intentional deviation from the reference,
  whose `decode_and_crop_jpeg` always decodes at full resolution before the bilinear resize
  (`imagenet_preprocessing.py:326-372`).  A crop of at least twice the output resolution is
  decoded by libjpeg at 1/2, 1/4 or 1/8 scale (box-filtered in the DCT domain) and then resized:
  the same picture content, slightly smoother pixels (mean absolute difference below 6 of 255 in
  `tests/test_data.py`), a quarter to a sixty-fourth of the decode work.  No reference fixture
  pins either result; `DTF_JPEG_DRAFT=0` selects the reference's full-size decode."""
  return os.environ.get("DTF_JPEG_DRAFT", "1") != "0"


def _decode_for_crop(image_buffer: bytes, rng, bbox: np.ndarray, out_h: int, out_w: int):
  """Training decode: samples the distorted crop box from the header's dimensions first, then
  lets libjpeg decode at 1/2, 1/4 or 1/8 scale (`Image.draft`: the inverse DCT produces the
  smaller image directly) when the crop still has at least the output resolution at that scale.
  An average ImageNet crop (0.08-1.0 of a ~400x350 image, resized to 224x224) rarely qualifies,
  large photographs always do: their decode time drops by the square of the scale.  Returns the
  pixels and the crop box in the decoded image's coordinates.  `DTF_JPEG_DRAFT=0` decodes at
  full size like the reference (`imagenet_preprocessing.py:259-323` crops while decoding but
  never scales)."""
  from PIL import Image  # pylint: disable=import-outside-toplevel
  img = Image.open(io.BytesIO(image_buffer))
  width, height = img.size
  y, x, h, w = sample_distorted_bounding_box(height, width, bbox, rng)
  if _jpeg_draft_enabled() and img.format == "JPEG" and h >= 2 * out_h and w >= 2 * out_w:
    # the full image at the size where the crop maps onto exactly the output resolution
    img.draft("RGB", (-(-width * out_w // w), -(-height * out_h // h)))
    sw, sh = img.size
    if (sw, sh) != (width, height):
      fx, fy = sw / width, sh / height
      x, w = int(x * fx), max(1, int(w * fx))
      y, h = int(y * fy), max(1, int(h * fy))
      w, h = min(w, sw - x), min(h, sh - y)
  if img.mode != "RGB":
    img = img.convert("RGB")
  return img, (y, x, h, w)


def _jpeg_region_decoder():
  """`_dtf_data` when its JPEG region decoder can be used (libjpeg-turbo found at run time,
  `DTF_JPEG_REGION` not 0), else None."""
  if os.environ.get("DTF_JPEG_REGION", "1") == "0":
    return None
  from . import native_loader  # pylint: disable=import-outside-toplevel
  if not native_loader.available():
    return None
  ext = native_loader.ext()
  return ext if getattr(ext, "jpeg_region_available", lambda: False)() else None


def _decode_region_for_crop(ext, image_buffer: bytes, rng, bbox: np.ndarray, out_h: int, out_w: int):
  """Training decode through the region decoder (`ops/csrc/jpeg_region.h`): the distorted crop
  box is sampled from the header's dimensions, then only the box's rows and MCU-aligned columns
  are decoded - what the reference's `tf.image.decode_and_crop_jpeg` does
  (`imagenet_preprocessing.py:259-323`) - at 1/2, 1/4 or 1/8 scale when the box still has the
  output resolution there.  Returns (uint8 pixels, box inside them) like `_decode_for_crop`, or
  None when the file is not a plain JPEG (CMYK, PNG content, damaged): the caller then decodes
  with PIL, drawing the same random numbers again."""
  hdr = ext.jpeg_header(image_buffer)
  if hdr is None or hdr[2] not in (1, 3):
    return None
  height, width = int(hdr[0]), int(hdr[1])
  gen = getattr(rng, "bit_generator", None)
  state = gen.state if gen is not None else None
  y, x, h, w = sample_distorted_bounding_box(height, width, bbox, rng)
  denom = 1
  if _jpeg_draft_enabled():
    while denom < 8 and h >= 2 * denom * out_h and w >= 2 * denom * out_w:
      denom *= 2
  # libjpeg's scaled image is ceil(size / denom) pixels; the box is mapped with that exact ratio
  # (floor of its first edge, ceiling of its last) so that it covers the same picture content
  sh, sw = -(-height // denom), -(-width // denom)
  ys, xs = y * sh // height, x * sw // width
  hs = max(1, min(sh, -(-(y + h) * sh // height)) - ys)
  ws = max(1, min(sw, -(-(x + w) * sw // width)) - xs)
  res = ext.jpeg_decode_region(image_buffer, ys, xs, hs, ws, denom)
  if res is None:
    if gen is not None:
      gen.state = state
    return None
  pixels, y0, x0, _sh, _sw = res
  by, bx = ys - y0, xs - x0
  return pixels, (by, bx, min(hs, pixels.shape[0] - by), min(ws, pixels.shape[1] - bx))


def _decode_crop_and_flip(image_buffer: bytes, bbox: np.ndarray, rng):
  """Random distorted-box crop + random horizontal flip. Returns a PIL image."""
  from PIL import Image  # pylint: disable=import-outside-toplevel
  img, (y, x, h, w) = _decode_for_crop(image_buffer, rng, bbox, DEFAULT_IMAGE_SIZE, DEFAULT_IMAGE_SIZE)
  img = img.crop((x, y, x + w, y + h))
  if rng.random() < 0.5:
    img = img.transpose(Image.FLIP_LEFT_RIGHT)
  return img


def _smallest_size_at_least(height: int, width: int, resize_min: int) -> Tuple[int, int]:
  """New (height, width) with the smaller side equal to `resize_min`, aspect kept."""
  smaller = float(min(height, width))
  scale = resize_min / smaller
  return int(height * scale), int(width * scale)


def _resize_image(img, height: int, width: int):
  from PIL import Image  # pylint: disable=import-outside-toplevel
  return img.resize((width, height), Image.BILINEAR)


def _aspect_preserving_resize(img, resize_min: int):
  width, height = img.size
  new_h, new_w = _smallest_size_at_least(height, width, resize_min)
  return _resize_image(img, new_h, new_w)


def _central_crop(image: np.ndarray, crop_height: int, crop_width: int) -> np.ndarray:
  h, w = image.shape[0], image.shape[1]
  top = (h - crop_height) // 2
  left = (w - crop_width) // 2
  return image[top:top + crop_height, left:left + crop_width]


def _mean_image_subtraction(image: np.ndarray, means, num_channels: int) -> np.ndarray:
  if image.ndim != 3:
    raise ValueError("Input must be of size [height, width, C>0]")
  if len(means) != num_channels:
    raise ValueError("len(means) must match the number of channels")
  return image - np.asarray(means, dtype=np.float32).reshape(1, 1, num_channels)


def _native_resize():
  """The C++ crop + bilinear resize (`_dtf_data.crop_resize_bilinear`), or None without the module."""
  from . import native_loader  # pylint: disable=import-outside-toplevel
  return native_loader.ext().crop_resize_bilinear if native_loader.available() else None


def preprocess_image(image_buffer: bytes, bbox: np.ndarray, output_height: int,
                     output_width: int, num_channels: int, is_training: bool = False,
                     rng=None) -> np.ndarray:
  """Decodes and preprocesses one image to `[output_height, output_width, C]` float32.

  Training: random distorted-box crop, random flip, bilinear resize; evaluation: aspect-
  preserving resize to a short side of 256 and a central crop (`imagenet_preprocessing.py:503-537`);
  then the channel means are subtracted.  With the native input module the crop, the resize
  (TensorFlow's legacy bilinear rule, which the reference's `_resize_image` uses), the flip and
  the mean subtraction are one C++ pass over the decoded image, and training JPEGs are decoded
  inside their crop box only; otherwise PIL decodes and resizes.
  """
  rng = rng or np.random.default_rng()
  fast = _native_resize() if num_channels == 3 else None
  if fast is not None:
    if is_training:
      region = _jpeg_region_decoder()
      got = _decode_region_for_crop(region, image_buffer, rng, bbox, output_height, output_width) \
          if region is not None else None
      if got is not None:
        pixels, (y, x, h, w) = got                         # uint8 [rows, cols, 3] around the box
      else:
        img, (y, x, h, w) = _decode_for_crop(image_buffer, rng, bbox, output_height, output_width)
        pixels = np.asarray(img)                           # uint8 [H, W, 3]
      flip = bool(rng.random() < 0.5)
      return fast(pixels, y, x, h, w, output_height, output_width, 0, 0, output_height,
                  output_width, flip, list(_CHANNEL_MEANS))
    region = _jpeg_region_decoder()
    hdr = region.jpeg_header(image_buffer) if region is not None else None
    if hdr is not None and hdr[2] in (1, 3):
      # evaluation: the central crop of the resized image needs the middle 7/8 of the source
      # only; rows and MCU-aligned columns outside it are not decoded
      height, width = int(hdr[0]), int(hdr[1])
      new_h, new_w = _smallest_size_at_least(height, width, _RESIZE_MIN)
      top, left = (new_h - output_height) // 2, (new_w - output_width) // 2
      sy, sx = np.float32(height) / np.float32(new_h), np.float32(width) / np.float32(new_w)
      y0 = min(int(np.float32(top) * sy), height - 1)
      y1 = min(int(np.float32(top + output_height - 1) * sy) + 1, height - 1)
      x0 = min(int(np.float32(left) * sx), width - 1)
      x1 = min(int(np.float32(left + output_width - 1) * sx) + 1, width - 1)
      res = region.jpeg_decode_region(image_buffer, y0, x0, y1 - y0 + 1, x1 - x0 + 1, 1)
      if res is not None:
        pixels, oy, ox, _ih, _iw = res
        return fast(pixels, 0, 0, height, width, new_h, new_w, top, left, output_height,
                    output_width, False, list(_CHANNEL_MEANS), oy, ox)
    pixels = np.asarray(_decode(image_buffer))           # uint8 [H, W, 3]
    height, width = pixels.shape[0], pixels.shape[1]
    new_h, new_w = _smallest_size_at_least(height, width, _RESIZE_MIN)
    top, left = (new_h - output_height) // 2, (new_w - output_width) // 2
    return fast(pixels, 0, 0, height, width, new_h, new_w, top, left, output_height,
                output_width, False, list(_CHANNEL_MEANS))
  if is_training:
    img = _decode_crop_and_flip(image_buffer, bbox, rng)
    img = _resize_image(img, output_height, output_width)
    image = np.asarray(img, dtype=np.float32)
  else:
    img = _decode(image_buffer)
    img = _aspect_preserving_resize(img, _RESIZE_MIN)
    image = _central_crop(np.asarray(img, dtype=np.float32), output_height, output_width)
  return _mean_image_subtraction(image, _CHANNEL_MEANS, num_channels)


def parse_record(raw_record: bytes, is_training: bool, dtype="fp32", rng=None):
  """Parses one serialized Example into (image HWC float32, label float32 [1])."""
  del dtype
  image_buffer, label, bbox = _parse_example_proto(raw_record)
  image = preprocess_image(image_buffer, bbox, DEFAULT_IMAGE_SIZE, DEFAULT_IMAGE_SIZE,
                           NUM_CHANNELS, is_training, rng)
  # Subtract one so that labels are in [0, 1000), float32 for the Keras-style model.
  label = np.array([label - 1], dtype=np.float32)
  return image.astype(np.float32), label


def input_fn(is_training, data_dir, batch_size, num_epochs=1, dtype="fp32",
             datasets_num_private_threads=None, parse_record_fn=parse_record,
             input_context=None, drop_remainder=False, tf_data_experimental_slack=False,
             device=None, shard_index=0, num_shards=1, filenames=None, **_):
  """Input function which provides batches for train or eval."""
  if filenames is None:
    filenames = get_filenames(is_training, data_dir)
  if input_context is not None:
    shard_index, num_shards = input_context
  if num_shards > 1:
    logging.info("Sharding the dataset: input_pipeline_id=%d num_input_pipelines=%d",
                 shard_index, num_shards)
  ds = pipeline.RecordDataset(
      filenames, read_tfrecords, parse_record_fn or parse_record, is_training,
      shuffle_buffer=_SHUFFLE_BUFFER, num_epochs=None if is_training else num_epochs,
      dtype=dtype, shard_index=shard_index, num_shards=num_shards, cycle_length=10)
  return pipeline.process_record_dataset(
      ds, is_training, batch_size, device=device,
      datasets_num_private_threads=datasets_num_private_threads,
      drop_remainder=drop_remainder, tf_data_experimental_slack=tf_data_experimental_slack)
