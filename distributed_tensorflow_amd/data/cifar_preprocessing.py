"""CIFAR-10 binary-record input pipeline.

Parity with the reference `image_classification/cifar_preprocessing.py`:
records of 1 label byte + 3072 image bytes in CHW order (`:27-40`); `parse_record`
converts to HWC float32 and a one-hot label (`:43-81`); training augmentation is
zero-pad to 40x40, random 32x32 crop, random horizontal flip, then per-image
standardisation for both training and eval (`:84-99`); `input_fn` reads
`data_batch_{1..5}.bin` / `test_batch.bin` with a 50 000-record shuffle buffer
(`:102-164`).
"""
from __future__ import annotations

import os
from typing import Iterator

import numpy as np

from ..utils import logging
from . import pipeline

HEIGHT = 32
WIDTH = 32
NUM_CHANNELS = 3
_DEFAULT_IMAGE_BYTES = HEIGHT * WIDTH * NUM_CHANNELS
# The record is the image plus a one-byte label
_RECORD_BYTES = _DEFAULT_IMAGE_BYTES + 1

NUM_IMAGES = {
    "train": 50000,
    "validation": 10000,
}
_NUM_DATA_FILES = 5
NUM_CLASSES = 10


def read_fixed_length_records(path: str, record_bytes: int = _RECORD_BYTES) -> Iterator[bytes]:
  """`FixedLengthRecordDataset` analogue."""
  with open(path, "rb") as f:
    while True:
      rec = f.read(record_bytes)
      if len(rec) < record_bytes:
        return
      yield rec


def per_image_standardization(image: np.ndarray) -> np.ndarray:
  """`(x - mean) / max(stddev, 1/sqrt(N))`, as `tf.image.per_image_standardization`."""
  image = image.astype(np.float32)
  mean = image.mean()
  std = max(float(image.std()), 1.0 / np.sqrt(image.size))
  return (image - mean) / std


def preprocess_image(image: np.ndarray, is_training: bool, rng=None) -> np.ndarray:
  """Preprocess a single HWC image."""
  if is_training:
    rng = rng or np.random.default_rng()
    padded = np.zeros((HEIGHT + 8, WIDTH + 8, NUM_CHANNELS), dtype=image.dtype)
    padded[4:4 + HEIGHT, 4:4 + WIDTH] = image
    y = int(rng.integers(0, 9))
    x = int(rng.integers(0, 9))
    image = padded[y:y + HEIGHT, x:x + WIDTH]
    if rng.random() < 0.5:
      image = image[:, ::-1]
  return per_image_standardization(image)


def parse_record(raw_record: bytes, is_training: bool, dtype="fp32", rng=None):
  """Parses a 3073-byte record into (HWC float32 image, one-hot float32 label)."""
  del dtype
  vec = np.frombuffer(raw_record, dtype=np.uint8)
  label = int(vec[0])
  depth_major = vec[1:_RECORD_BYTES].reshape(NUM_CHANNELS, HEIGHT, WIDTH)
  image = depth_major.transpose(1, 2, 0).astype(np.float32)
  image = preprocess_image(image, is_training, rng)
  onehot = np.zeros((NUM_CLASSES,), dtype=np.float32)
  onehot[label] = 1.0
  return image, onehot


def torch_device_type(device) -> str:
  import torch  # pylint: disable=import-outside-toplevel
  return torch.device(device).type


def get_filenames(is_training: bool, data_dir: str):
  """Returns a list of filenames."""
  assert os.path.exists(data_dir), (
      "Download and extract the CIFAR-10 binary data (cifar-10-batches-bin) into data_dir; "
      "missing: %s" % data_dir)
  if is_training:
    return [os.path.join(data_dir, "data_batch_%d.bin" % i)
            for i in range(1, _NUM_DATA_FILES + 1)]
  return [os.path.join(data_dir, "test_batch.bin")]


def input_fn(is_training, data_dir, batch_size, num_epochs=1, dtype="fp32",
             datasets_num_private_threads=None, parse_record_fn=parse_record,
             input_context=None, drop_remainder=False, device=None, shard_index=0,
             num_shards=1, tf_data_experimental_slack=False, **_):
  """Input function which provides batches for train or eval."""
  filenames = get_filenames(is_training, data_dir)
  if input_context is not None:
    shard_index, num_shards = input_context
  if num_shards > 1:
    logging.info("Sharding the dataset: input_pipeline_id=%d num_input_pipelines=%d",
                 shard_index, num_shards)
  from . import native_loader  # pylint: disable=import-outside-toplevel
  if (parse_record_fn is None or parse_record_fn is parse_record) and native_loader.available():
    # C++ pipeline: same records, same augmentation, a thread pool instead of worker processes
    on_gpu = device is not None and torch_device_type(device) == "cuda"
    ds = native_loader.NativeCifarDataset(
        filenames, is_training, batch_size, num_epochs=None if is_training else num_epochs,
        shard_index=shard_index, num_shards=num_shards,
        num_threads=datasets_num_private_threads, drop_remainder=drop_remainder,
        pin_memory=on_gpu, ring=6 if tf_data_experimental_slack else 4)
    delay = float(os.environ.get("DTF_DATA_PREFETCH_DELAY_MS", "0"))
    pf = pipeline.DevicePrefetcher(ds, device if device is not None else "cpu",
                                   depth=3 if tf_data_experimental_slack else 2, delay_ms=delay)
    pf.batch_size = batch_size
    return pf
  ds = pipeline.RecordDataset(
      filenames, read_fixed_length_records, parse_record_fn or parse_record, is_training,
      shuffle_buffer=NUM_IMAGES["train"], num_epochs=None if is_training else num_epochs,
      dtype=dtype, shard_index=shard_index, num_shards=num_shards, shard_records=True)
  return pipeline.process_record_dataset(
      ds, is_training, batch_size, device=device,
      datasets_num_private_threads=datasets_num_private_threads,
      drop_remainder=drop_remainder, tf_data_experimental_slack=tf_data_experimental_slack)
