"""Python face of the C++ input pipeline (`ops/_dtf_data`).

`NativeCifarDataset` yields `(images [B,32,32,3] float32, labels [B,10] float32)` batches
assembled by a C++ thread pool (decode CHW bytes, pad / random crop / random flip, per-image
standardisation: the map function of `cifar_preprocessing.py:84-99` in the reference) into a
ring of pinned host buffers, so that the only Python work per batch is one function call.
It plugs into `pipeline.DevicePrefetcher` like a `DataLoader`.
"""
from __future__ import annotations

import importlib
import os
from typing import Optional, Sequence

import torch

__all__ = ["available", "ext", "NativeCifarDataset", "read_tfrecords"]

_ext = None
_err: Optional[Exception] = None
try:
  _ext = importlib.import_module("distributed_tensorflow_amd.ops._dtf_data")
except Exception as e:  # pragma: no cover - depends on the build
  _err = e


def available() -> bool:
  return _ext is not None and os.environ.get("DTF_NATIVE_DATA", "1") != "0"


def ext():
  if _ext is None:
    raise RuntimeError("native data module _dtf_data is not built; run "
                       "`python -m distributed_tensorflow_amd.ops.build` (%s)" % (_err,))
  return _ext


def read_tfrecords(path: str, verify: bool = True):
  """All record payloads of a TFRecord file, checksums verified in C++."""
  return ext().tfrecord_read_all(path, verify)


class NativeCifarDataset:
  """Batched CIFAR-10 iterator backed by `_dtf_data.CifarLoader`."""

  def __init__(self, filenames: Sequence[str], is_training: bool, batch_size: int,
               num_epochs: Optional[int] = None, shard_index: int = 0, num_shards: int = 1,
               seed: int = 0, num_threads: Optional[int] = None, drop_remainder: bool = False,
               pin_memory: bool = False, ring: int = 4):
    if num_threads is None:
      num_threads = min(max((os.cpu_count() or 2) // 2, 1), 8)
    self.batch_size = int(batch_size)
    self.is_training = is_training
    self.drop_remainder = drop_remainder
    self.num_epochs = 0 if num_epochs is None else int(num_epochs)     # 0: repeat forever
    self._args = (list(filenames), bool(is_training), self.batch_size, int(shard_index),
                  int(num_shards), int(seed), int(num_threads), bool(drop_remainder),
                  self.num_epochs)
    self._loader = ext().CifarLoader(*self._args)
    self.num_records = self._loader.num_records()
    # a ring of host buffers: a batch handed out stays valid while the next ones are filled
    mk = lambda *shape: (torch.empty(*shape).pin_memory() if pin_memory else torch.empty(*shape))
    self._ring = [(mk(self.batch_size, 32, 32, 3), mk(self.batch_size, 10))
                  for _ in range(max(ring, 2))]
    self._slot = 0

  def __len__(self):
    per_epoch = self.num_records // self.batch_size if self.drop_remainder else \
        -(-self.num_records // self.batch_size)
    return per_epoch * max(self.num_epochs, 1)

  def __iter__(self):
    if not self.is_training:
      self._loader = ext().CifarLoader(*self._args)     # every evaluation starts at record 0
    while True:
      images, labels = self._ring[self._slot]
      self._slot = (self._slot + 1) % len(self._ring)
      n = self._loader.next_batch(images.numpy(), labels.numpy())
      if n == 0:
        return
      yield images[:n], labels[:n]
