"""Training-loop callbacks (Keras callback protocol) and the reference's observability.

* `TimeHistory` / `BatchTimestamp`: per-`log_steps` timestamps and the
  `BenchmarkMetric: {...}` log lines whose format is fixed by the reference logs
  (`ps_server/log1.log:134-142`); contract pinned by `common_test.py:68-88`
  (7 batches with `log_steps=3` give exactly 3 timestamps).
* `LearningRateBatchScheduler` (`common.py:36-73`): host-side per-batch schedule
  that only touches the optimizer when the value changes.
* `get_profiler_callback` (`common.py:192-197`): `--profile_steps a,b` window.
* `TensorBoard` (`common.py:187-190`): scalar event files in `model_dir`.
* `ModelCheckpoint` (`resnet_cifar_main_horovod.py:235-237`): full-model HDF5 per epoch.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional

import numpy as np

from ..utils import logging
from ..utils import tfrecord

__all__ = ["StopAtThreshold", "Callback", "CallbackList", "History", "BatchTimestamp", "TimeHistory",
           "LearningRateBatchScheduler", "ProfilerCallback", "get_profiler_callback",
           "TensorBoard", "ModelCheckpoint"]


class Callback:
  """Keras callback protocol (the subset the reference's callbacks use)."""

  def __init__(self):
    self.model = None
    self.params: Dict = {}

  def set_model(self, model):
    self.model = model

  def set_params(self, params):
    self.params = params

  def on_train_begin(self, logs=None): ...
  def on_train_end(self, logs=None): ...
  def on_epoch_begin(self, epoch, logs=None): ...
  def on_epoch_end(self, epoch, logs=None): ...
  def on_batch_begin(self, batch, logs=None): ...
  def on_batch_end(self, batch, logs=None): ...
  def on_test_begin(self, logs=None): ...
  def on_test_end(self, logs=None): ...

  # Keras >= 2.3 spellings
  def on_train_batch_begin(self, batch, logs=None):
    self.on_batch_begin(batch, logs)

  def on_train_batch_end(self, batch, logs=None):
    self.on_batch_end(batch, logs)


class CallbackList:
  def __init__(self, callbacks: Optional[List[Callback]], model, params=None):
    self.callbacks = list(callbacks or [])
    for c in self.callbacks:
      c.set_model(model)
      c.set_params(params or {})

  def __iter__(self):
    return iter(self.callbacks)

  def call(self, hook: str, *args):
    for c in self.callbacks:
      getattr(c, hook)(*args)

  @property
  def wants_batch_logs(self) -> bool:
    """True when some callback reads per-batch metric values (forces a device sync)."""
    return any(getattr(c, "needs_batch_logs", False) for c in self.callbacks)


class History(Callback):
  """Records epoch-level metrics into `history` (returned by `Model.fit`)."""

  def __init__(self):
    super().__init__()
    self.history: Dict[str, list] = {}
    self.epoch: List[int] = []

  def on_epoch_end(self, epoch, logs=None):
    self.epoch.append(epoch)
    for k, v in (logs or {}).items():
      self.history.setdefault(k, []).append(v)


class BatchTimestamp:
  """A structure to store batch time stamp."""

  def __init__(self, batch_index, timestamp):
    self.batch_index = batch_index
    self.timestamp = timestamp

  def __repr__(self):
    return "'BatchTimestamp<batch_index: {}, timestamp: {}>'".format(
        self.batch_index, self.timestamp)


class TimeHistory(Callback):
  """Throughput log: a timestamp every `log_steps` global steps."""

  def __init__(self, batch_size, log_steps, synchronize: Optional[Callable[[], None]] = None):
    """`batch_size` is the global batch size; `synchronize` (optional) drains the
    device before a timestamp is taken so that asynchronous kernel launches do
    not make the reported examples/second optimistic."""
    super().__init__()
    self.batch_size = batch_size
    self.log_steps = log_steps
    self.global_steps = 0
    self.timestamp_log: List[BatchTimestamp] = []
    self.train_finish_time = None
    self.start_time = None
    self.epoch_start = None
    self._sync = synchronize

  def _now(self):
    if self._sync is not None:
      self._sync()
    elif self.model is not None and hasattr(self.model, "synchronize"):
      self.model.synchronize()
    return time.time()

  def on_train_end(self, logs=None):
    self.train_finish_time = self._now()

  def on_epoch_begin(self, epoch, logs=None):
    self.epoch_start = time.time()

  def on_batch_begin(self, batch, logs=None):
    self.global_steps += 1
    if self.global_steps == 1:
      self.start_time = self._now()
      self.timestamp_log.append(BatchTimestamp(self.global_steps, self.start_time))

  def on_batch_end(self, batch, logs=None):
    """Records elapse time of the batch and calculates examples per second."""
    if self.global_steps % self.log_steps == 0:
      timestamp = self._now()
      elapsed_time = timestamp - self.start_time
      examples_per_second = (self.batch_size * self.log_steps) / elapsed_time
      self.timestamp_log.append(BatchTimestamp(self.global_steps, timestamp))
      logging.info(
          "BenchmarkMetric: {'global step':%d, 'time_taken': %f,"
          "'examples_per_second': %f}", self.global_steps, elapsed_time, examples_per_second)
      self.start_time = timestamp

  def on_epoch_end(self, epoch, logs=None):
    epoch_run_time = self._now() - self.epoch_start
    logging.info("BenchmarkMetric: {'epoch':%d, 'time_taken': %f}", epoch, epoch_run_time)


class LearningRateBatchScheduler(Callback):
  """Callback to update learning rate on every batch (not epoch boundaries).

  `schedule(epoch, batch, batches_per_epoch, batch_size) -> float`, both indices
  from 0.  The optimizer is only written when the value changed.
  """

  def __init__(self, schedule, batch_size, num_images):
    super().__init__()
    self.schedule = schedule
    self.batches_per_epoch = num_images / batch_size
    self.batch_size = batch_size
    self.epochs = -1
    self.prev_lr = -1

  def on_epoch_begin(self, epoch, logs=None):
    if not hasattr(self.model.optimizer, "learning_rate"):
      raise ValueError('Optimizer must have a "learning_rate" attribute.')
    self.epochs += 1

  def on_batch_begin(self, batch, logs=None):
    lr = self.schedule(self.epochs, batch, self.batches_per_epoch, self.batch_size)
    if not isinstance(lr, (float, np.float32, np.float64)):
      raise ValueError('The output of the "schedule" function should be float.')
    if lr != self.prev_lr:
      self.model.optimizer.learning_rate = lr
      self.prev_lr = lr
      logging.debug("Epoch %05d Batch %05d: LearningRateBatchScheduler change "
                    "learning rate to %s.", self.epochs, batch, lr)


class ProfilerCallback(Callback):
  """Captures a kernel/host trace for global steps `start..stop` into `log_dir`.

  Uses `torch.profiler` (roctracer underneath on ROCm): the trace is written as a
  Chrome/Perfetto JSON plus a per-kernel time table, which is the information
  the reference obtained from the TF profiler.
  """

  def __init__(self, log_dir, start_step, stop_step):
    super().__init__()
    self.log_dir = log_dir
    self.start_step = start_step
    self.stop_step = stop_step
    self._prof = None
    self._step = 0

  def on_batch_begin(self, batch, logs=None):
    self._step += 1
    if self._step == self.start_step:
      import torch  # pylint: disable=import-outside-toplevel
      acts = [torch.profiler.ProfilerActivity.CPU]
      if torch.cuda.is_available():
        acts.append(torch.profiler.ProfilerActivity.CUDA)
      self._prof = torch.profiler.profile(activities=acts)
      self._prof.__enter__()
      logging.info("Profiler started at Step %s", self.start_step)

  def on_batch_end(self, batch, logs=None):
    if self._step == self.stop_step and self._prof is not None:
      if self.model is not None:
        self.model.synchronize()
      self._prof.__exit__(None, None, None)
      os.makedirs(self.log_dir, exist_ok=True)
      path = os.path.join(self.log_dir, "profile_steps_%d_%d.trace.json"
                          % (self.start_step, self.stop_step))
      self._prof.export_chrome_trace(path)
      try:
        table = self._prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40)
        with open(os.path.join(self.log_dir, "profile_steps_%d_%d.txt"
                               % (self.start_step, self.stop_step)), "w") as f:
          f.write(table)
      except Exception as e:  # pragma: no cover
        logging.warning("could not write profiler table: %s", e)
      logging.info("Profiler saved profiles for steps between %s and %s to %s",
                   self.start_step, self.stop_step, self.log_dir)
      self._prof = None


def get_profiler_callback(model_dir, profile_steps, enable_tensorboard=False):
  """Validates `--profile_steps "start,stop"` and returns the callback."""
  del enable_tensorboard
  profile_steps_error_message = (
      "profile_steps must be a comma separated pair of positive integers, "
      "specifying the first and last steps to be profiled.")
  try:
    profile_steps = [int(i) for i in profile_steps.split(",")]
  except ValueError:
    raise ValueError(profile_steps_error_message) from None
  if len(profile_steps) != 2:
    raise ValueError(profile_steps_error_message)
  start_step, stop_step = profile_steps
  if start_step < 0 or start_step > stop_step:
    raise ValueError(profile_steps_error_message)
  return ProfilerCallback(model_dir, start_step, stop_step)


class TensorBoard(Callback):
  """Writes epoch scalars (and the learning rate) as TensorBoard event files."""

  def __init__(self, log_dir):
    super().__init__()
    self.log_dir = log_dir
    self._f = None

  def _open(self):
    if self._f is None:
      d = os.path.join(self.log_dir, "train")
      os.makedirs(d, exist_ok=True)
      fname = "events.out.tfevents.%d.%s" % (int(time.time()), os.uname().nodename)
      self._f = open(os.path.join(d, fname), "wb")
      tfrecord.write_record(self._f, tfrecord.encode_event_file_version(time.time()))

  def on_train_begin(self, logs=None):
    self._open()

  def on_epoch_end(self, epoch, logs=None):
    self._open()
    now = time.time()
    for k, v in (logs or {}).items():
      tag = ("epoch_" + k) if not k.startswith("val_") else ("val/epoch_" + k[4:])
      tfrecord.write_record(self._f, tfrecord.encode_event_scalar(now, epoch, tag, float(v)))
    if self.model is not None and getattr(self.model, "optimizer", None) is not None:
      tfrecord.write_record(self._f, tfrecord.encode_event_scalar(
          now, epoch, "epoch_learning_rate", self.model.optimizer.current_lr()))
    self._f.flush()

  def on_train_end(self, logs=None):
    if self._f is not None:
      self._f.close()
      self._f = None


class StopAtThreshold(Callback):
  """`--stop_threshold`: ends training once the evaluation accuracy reaches the threshold
  (`model_helpers.past_stop_threshold` applied to the validation metric of an epoch)."""

  def __init__(self, threshold, metric: str):
    super().__init__()
    self.threshold = threshold
    self.metric = metric

  def on_epoch_end(self, epoch, logs=None):
    from ..utils import model_helpers  # pylint: disable=import-outside-toplevel
    val = (logs or {}).get("val_" + self.metric)
    if val is not None and model_helpers.past_stop_threshold(self.threshold, val):
      self.model.stop_training = True


class ModelCheckpoint(Callback):
  """Saves the full model (weights + optimizer slots) after every epoch.

  `filepath` may contain `{epoch}` (1-based, as Keras formats it) and metric names.

  Under a synchronous strategy with several replicas the callback has to be installed on every
  replica: saving is collective there (the BatchNorm moving statistics written are the replica
  mean, see `Model.save_replicated`) and only the chief writes the file.  The Horovod drivers run
  without a strategy object and install it on rank 0 only, which saves rank 0's copy as the
  reference does (`resnet_imagenet_main_horovod.py:258-259`).
  """

  def __init__(self, filepath, save_weights_only=False, verbose=0):
    super().__init__()
    self.filepath = filepath
    self.save_weights_only = save_weights_only
    self.verbose = verbose

  def on_epoch_end(self, epoch, logs=None):
    path = self.filepath.format(epoch=epoch + 1, **(logs or {}))
    wrote = self.model.save_replicated(path, weights_only=self.save_weights_only)
    if self.verbose and wrote:
      logging.info("Epoch %05d: saving model to %s", epoch + 1, path)
