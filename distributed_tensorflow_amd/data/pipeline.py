"""Record-dataset engine: sharded files -> shuffle -> parallel parse -> batch -> device prefetch.

Plays the role of the reference's `process_record_dataset`
(`imagenet_preprocessing.py:65-141`): private worker pool
(`datasets_num_private_threads`), record shuffle buffer, `repeat(num_epochs)`,
parallel `parse_record_fn`, `batch(drop_remainder)`, and a final prefetch that
keeps host work off the training critical path.  Worker *processes* (one intra-op
thread each, as the reference sets `max_intra_op_parallelism = 1`) read disjoint
file subsets; batches are assembled in pinned memory and copied to the GPU on a
side HIP stream one step ahead of the consumer.
"""
from __future__ import annotations

import os
import random
import threading
import queue
import time
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, IterableDataset, get_worker_info

__all__ = ["RecordDataset", "DevicePrefetcher", "process_record_dataset"]


class RecordDataset(IterableDataset):
  """Iterable over parsed `(image, label)` samples of a sharded record file set.

  `record_reader(path) -> iterator of raw records`; `parse_record_fn(raw,
  is_training, dtype) -> (image ndarray HWC float32, label ndarray)`.
  """

  def __init__(self, filenames: Sequence[str], record_reader: Callable[[str], Iterator[bytes]],
               parse_record_fn: Callable, is_training: bool, shuffle_buffer: int,
               num_epochs: Optional[int] = 1, dtype="fp32", shard_index: int = 0,
               num_shards: int = 1, shard_records: bool = False, seed: int = 0,
               cycle_length: int = 10):
    super().__init__()
    self.filenames = list(filenames)
    self.record_reader = record_reader
    self.parse_record_fn = parse_record_fn
    self.is_training = is_training
    self.shuffle_buffer = shuffle_buffer if is_training else 0
    self.num_epochs = num_epochs
    self.dtype = dtype
    self.shard_index = shard_index
    self.num_shards = max(num_shards, 1)
    self.shard_records = shard_records   # few files: shard by record index instead of by file
    self.seed = seed
    self.cycle_length = cycle_length

  def _my_files(self) -> Tuple[List[str], int, int]:
    """Files of this (rank, worker) and the record-level (stride, offset) if any."""
    wi = get_worker_info()
    nw, wid = (wi.num_workers, wi.id) if wi is not None else (1, 0)
    total = self.num_shards * nw
    me = self.shard_index * nw + wid
    if not self.shard_records and len(self.filenames) >= total:
      return self.filenames[me::total], 1, 0
    return list(self.filenames), total, me

  def _raw_records(self, files: List[str], rng: random.Random) -> Iterator[bytes]:
    """Interleaves up to `cycle_length` files (reference: interleave cycle_length=10)."""
    files = list(files)
    if self.is_training:
      rng.shuffle(files)
    pending = iter(files)
    active = []
    for _ in range(self.cycle_length):
      f = next(pending, None)
      if f is None:
        break
      active.append(self.record_reader(f))
    while active:
      nxt = []
      for it in active:
        rec = next(it, None)
        if rec is None:
          f = next(pending, None)
          if f is not None:
            it2 = self.record_reader(f)
            rec = next(it2, None)
            if rec is not None:
              nxt.append(it2)
              yield rec
          continue
        nxt.append(it)
        yield rec
      active = nxt

  def __iter__(self):
    torch.set_num_threads(1)
    wi = get_worker_info()
    wid = wi.id if wi is not None else 0
    rng = random.Random(self.seed * 7919 + self.shard_index * 131 + wid)
    np_rng = np.random.default_rng(self.seed * 7919 + self.shard_index * 131 + wid)
    files, stride, offset = self._my_files()
    epoch = 0
    while self.num_epochs is None or epoch < self.num_epochs:
      epoch += 1
      buf: List[bytes] = []
      for i, rec in enumerate(self._raw_records(files, rng)):
        if stride > 1 and i % stride != offset:
          continue
        if self.shuffle_buffer > 1:
          if len(buf) < self.shuffle_buffer:
            buf.append(rec)
            continue
          j = rng.randrange(len(buf))
          rec, buf[j] = buf[j], rec
        yield self._parse(rec, np_rng)
      rng.shuffle(buf)
      for rec in buf:
        yield self._parse(rec, np_rng)

  _TRANSPORT = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16,
                "float16": torch.float16}

  def _parse(self, rec, np_rng):
    image, label = self.parse_record_fn(rec, self.is_training, self.dtype, rng=np_rng)
    image = torch.from_numpy(np.ascontiguousarray(image))
    # The reference casts the image to the compute dtype inside the input pipeline
    # (`imagenet_preprocessing.py:250-255`).  Doing it here, in the worker, halves what the
    # pinned-memory thread and the host-to-device copy have to move per batch (fp32 batches of
    # 256 x 224 x 224 x 3 capped the pipeline near 8k images/s).
    t = self._TRANSPORT.get(str(self.dtype))
    if t is not None:
      image = image.to(t)
    return image, torch.from_numpy(np.asarray(label))


class _Rebatcher:
  """Iterable over the batches of a multi-worker `DataLoader` with batch boundaries restored.

  `batch_size` and `drop_last` of a `DataLoader` over an `IterableDataset` apply *per worker*:
  each worker ends its stream with a partial batch of its own.  Dropping those (drop_last)
  loses up to `workers * (batch_size - 1)` samples per pass - enough to run an evaluation of
  `num_images // batch_size` steps out of data - and keeping them yields short batches in the
  middle of the stream.  The workers therefore never drop anything; full batches pass through
  untouched and partial ones are merged here, so that only the very last batch of the whole
  stream can be short (and is dropped when `drop_remainder` is set)."""

  def __init__(self, loader, batch_size: int, drop_remainder: bool):
    self.loader = loader
    self.batch_size = int(batch_size)
    self.drop_remainder = drop_remainder
    self.num_samples: Optional[int] = None     # set by whoever knows the stream's sample count

  def __len__(self):
    if self.num_samples is None:
      raise TypeError("this input stream has no known length (its sample count was not given): "
                      "pass `steps` / `steps_per_epoch` explicitly")
    n, bs = self.num_samples, self.batch_size
    return n // bs if self.drop_remainder else -(-n // bs)

  def __iter__(self):
    bs = self.batch_size
    carry_x, carry_y, have = [], [], 0
    for images, labels in self.loader:
      n = int(images.shape[0])
      if n == bs and not have:
        yield images, labels
        continue
      carry_x.append(images)
      carry_y.append(labels)
      have += n
      while have >= bs:
        x = torch.cat(carry_x) if len(carry_x) > 1 else carry_x[0]
        y = torch.cat(carry_y) if len(carry_y) > 1 else carry_y[0]
        yield x[:bs], y[:bs]
        have -= bs
        carry_x, carry_y = ([x[bs:]], [y[bs:]]) if have else ([], [])
    if have and not self.drop_remainder:
      yield (torch.cat(carry_x) if len(carry_x) > 1 else carry_x[0],
             torch.cat(carry_y) if len(carry_y) > 1 else carry_y[0])


class DevicePrefetcher:
  """Copies batches to the device one step ahead on a side stream (final `prefetch`)."""

  def __init__(self, loader, device: torch.device, depth: int = 2, delay_ms: float = 0.0):
    self.loader = loader
    self.device = torch.device(device)
    self.depth = max(depth, 1)
    self.delay_ms = delay_ms
    self.batch_size = getattr(loader, "batch_size", None)

  def __len__(self):
    return len(self.loader)

  def __iter__(self):
    # starts the worker processes; they come from a fork server that never touched the GPU
    # (process_record_dataset), not from a fork of this process
    it = iter(self.loader)
    if self.device.type != "cuda":
      for images, labels in it:
        yield images, labels
      return
    stream = torch.cuda.Stream(self.device)
    q: "queue.Queue" = queue.Queue(maxsize=self.depth)
    stop = threading.Event()

    def producer():
      try:
        for images, labels in it:
          if stop.is_set():
            return
          if self.delay_ms:
            time.sleep(self.delay_ms / 1000.0)
          with torch.cuda.stream(stream):
            im = images.to(self.device, non_blocking=True)
            lb = labels.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
          # The source may be a pinned staging buffer that its owner refills a few batches
          # later (the native loaders hand out slots of a small ring): the copy must have
          # read it before the next batch is requested.  This thread has nothing else to do
          # meanwhile, and the consumer still only waits on the device side.
          ev.synchronize()
          q.put((im, lb, ev))
      except BaseException as e:  # pylint: disable=broad-except
        q.put(e)                  # decode errors, CRC mismatches, a dead worker: raised in the
        return                    # consumer instead of looking like the end of the data
      q.put(None)

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    try:
      while True:
        item = q.get()
        if item is None:
          return
        if isinstance(item, BaseException):
          raise RuntimeError("the input pipeline failed: %r" % (item,)) from item
        im, lb, ev = item
        torch.cuda.current_stream(self.device).wait_event(ev)
        im.record_stream(torch.cuda.current_stream(self.device))
        lb.record_stream(torch.cuda.current_stream(self.device))
        yield im, lb
    finally:
      stop.set()


def process_record_dataset(dataset: RecordDataset, is_training: bool, batch_size: int,
                           device=None, datasets_num_private_threads: Optional[int] = None,
                           drop_remainder: bool = False, tf_data_experimental_slack: bool = False,
                           num_samples: Optional[int] = None):
  """Batches `dataset` with a private worker pool and prefetches to `device`."""
  del is_training
  workers = datasets_num_private_threads
  if workers is None:
    try:
      avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - platforms without affinity masks
      avail = os.cpu_count() or 2
    # ~2 ms of JPEG decoding per image and worker: 16 workers keep up with ~8k images/s
    workers = min(max(avail // 2, 1), 16)
  workers = max(int(workers), 0)
  if workers > len(dataset.filenames) and not dataset.shard_records:
    dataset.shard_records = True
  pin = device is not None and torch.device(device).type == "cuda"
  # Worker processes must not be forked from a process that has initialised the GPU (a HIP
  # call, or a destructor of an inherited GPU object, aborts the child): they are started by a
  # fork server, a helper process created from a clean interpreter that forks the workers on
  # request.  CPU-only runs keep the plain fork (cheaper, nothing to inherit).
  ctx = None
  if workers and (pin or torch.cuda.is_initialized()):
    ctx = os.environ.get("DTF_DATA_WORKER_START", "forkserver")
  loader = DataLoader(dataset, batch_size=batch_size, drop_last=False,
                      num_workers=workers, pin_memory=pin,
                      prefetch_factor=(4 if tf_data_experimental_slack else 2) if workers else None,
                      persistent_workers=False, multiprocessing_context=ctx)
  loader = _Rebatcher(loader, batch_size, drop_remainder)
  delay = float(os.environ.get("DTF_DATA_PREFETCH_DELAY_MS", "0"))
  pf = DevicePrefetcher(loader, device or torch.device("cpu"),
                        depth=3 if tf_data_experimental_slack else 2, delay_ms=delay)
  pf.batch_size = batch_size
  loader.num_samples = num_samples     # DevicePrefetcher.__len__ asks the loader
  return pf
