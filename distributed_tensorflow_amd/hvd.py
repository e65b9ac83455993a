"""Horovod-style API on the framework's communicator.

The reference's Horovod drivers (`resnet_cifar_main_horovod.py`,
`resnet_imagenet_main_horovod.py`) use exactly this surface of
`horovod.tensorflow.keras`:

    hvd.init(); hvd.rank(); hvd.local_rank(); hvd.size()
    hvd.DistributedOptimizer(opt)
    hvd.callbacks.BroadcastGlobalVariablesCallback(0)
    hvd.callbacks.MetricAverageCallback()
    hvd.callbacks.LearningRateWarmupCallback(warmup_epochs=3, verbose=1)

Launch one process per GPU with `python -m distributed_tensorflow_amd.launch -np N`
(the `horovodrun` / `mpirun` replacement); `init()` joins the RCCL (or gloo)
process group described by the launcher environment.

`DistributedOptimizer` averages gradients with *tensor fusion*: the gradients
already live in one flat buffer, cut into buckets of `fusion_threshold` bytes
(Horovod's default 64 MB); each bucket is all-reduced as soon as backward has
produced it, overlapped with the rest of backward.
"""
from __future__ import annotations

import types
from typing import Optional

import numpy as np
import torch

from .parallel.allreduce import GradientReducer
from .parallel.cluster import resolve_task
from .parallel.comm import Communicator, get_communicator
from .parallel import comm as _comm_mod
from .train.callbacks import Callback
from .train.optimizers import Optimizer
from .utils import logging

__all__ = ["init", "shutdown", "rank", "local_rank", "size", "local_size", "is_initialized",
           "allreduce", "broadcast", "broadcast_variables", "DistributedOptimizer",
           "callbacks", "device"]

_state = {"comm": None}

DEFAULT_FUSION_THRESHOLD = 64 * 1024 * 1024   # bytes; Horovod's HOROVOD_FUSION_THRESHOLD default


def init(use_gpu: Optional[bool] = None) -> Communicator:
  if _state["comm"] is None:
    task = resolve_task()
    if use_gpu is None:
      use_gpu = torch.cuda.is_available()
    _state["comm"] = get_communicator(task, use_gpu=use_gpu)
    _publish_host(_state["comm"])
  return _state["comm"]


def _publish_host(c: Communicator):
  """Every rank calls `init()`: this is where the ranks exchange their host names (through the
  job's key-value store), so that `local_size()` is a purely local query afterwards, as in
  Horovod, and never a hidden collective."""
  import os  # pylint: disable=import-outside-toplevel
  if "LOCAL_WORLD_SIZE" in os.environ:
    _state["local_size"] = int(os.environ["LOCAL_WORLD_SIZE"])
    return
  if c.world_size <= 1:
    _state["local_size"] = 1
    return
  import socket  # pylint: disable=import-outside-toplevel
  me = socket.gethostname()
  store = c.store()
  store.set("hvd/host/%d" % c.rank, me)
  hosts = [store.get("hvd/host/%d" % r).decode() for r in range(c.world_size)]   # blocks
  _state["local_size"] = sum(1 for h in hosts if h == me)


def shutdown():
  _state["comm"] = None
  _state.pop("local_size", None)
  _comm_mod.shutdown()


def is_initialized() -> bool:
  return _state["comm"] is not None


def _c() -> Communicator:
  if _state["comm"] is None:
    raise ValueError("Horovod has not been initialized; use hvd.init().")
  return _state["comm"]


def rank() -> int:
  return _c().rank


def local_rank() -> int:
  return _c().task.local_rank


def size() -> int:
  return _c().world_size


def local_size() -> int:
  """Ranks of this job on this host (`hvd.local_size()`): a local query.

  Launchers that spawn the ranks of a host export LOCAL_WORLD_SIZE; without it (ranks started
  by hand from a TF_CONFIG, one per terminal or host) the ranks exchanged their host names once
  inside `init()` and the count of matches was cached there."""
  _c()
  return _state["local_size"]


def device() -> torch.device:
  """The device this rank is pinned to (`set_visible_devices(gpus[local_rank])` analogue)."""
  return _c().device


def allreduce(tensor: torch.Tensor, average: bool = True, name: Optional[str] = None):
  """Returns the sum (or average) of `tensor` over all ranks (input left untouched)."""
  del name
  out = tensor.clone()
  _c().all_reduce(out, average=average)
  return out


def broadcast(tensor: torch.Tensor, root_rank: int = 0, name: Optional[str] = None):
  del name
  out = tensor.clone()
  _c().broadcast(out, root_rank)
  return out


def broadcast_variables(store, root_rank: int = 0):
  """In-place broadcast of the model variables and optimizer slots from `root_rank`."""
  c = _c()
  c.broadcast(store.params, root_rank)
  c.broadcast(store.state, root_rank)
  for name in sorted(store.slots):
    c.broadcast(store.slots[name], root_rank)
  store.version += 1


class DistributedOptimizer(Optimizer):
  """Wraps an optimizer: fused all-reduce *average* of the gradients, then apply."""

  def __init__(self, optimizer: Optimizer, name: Optional[str] = None,
               fusion_threshold: int = DEFAULT_FUSION_THRESHOLD,
               all_reduce_alg: Optional[str] = None):
    del name
    self.inner = optimizer
    self.fusion_bytes = int(fusion_threshold)
    self.all_reduce_alg = all_reduce_alg

  @property
  def learning_rate(self):
    return self.inner.learning_rate

  @learning_rate.setter
  def learning_rate(self, v):
    self.inner.learning_rate = v

  lr = learning_rate

  @property
  def iterations(self):
    return self.inner.iterations

  @iterations.setter
  def iterations(self, v):
    self.inner.iterations = v

  @property
  def slot_names(self):
    return self.inner.slot_names

  @property
  def momentum(self):
    return getattr(self.inner, "momentum", 0.0)

  def current_lr(self):
    return self.inner.current_lr()

  def make_reducer(self, store, buckets):
    c = _c()
    if c.world_size <= 1:
      return None
    return GradientReducer(c, store, buckets, self.all_reduce_alg)

  def apply_gradients(self, store, grad_divisor: float = 1.0, executor=None):
    # The Model has already run the reducer (sum); grad_divisor == size() turns it
    # into Horovod's average.
    return self.inner.apply_gradients(store, grad_divisor, executor)

  def get_config(self):
    return self.inner.get_config()


# ---------------------------------------------------------------------------------------
class BroadcastGlobalVariablesCallback(Callback):
  """Broadcast all variables (and optimizer slots) from `root_rank` at train begin.

  Needed for consistent initialisation when training starts from random weights
  or from a checkpoint restored on one rank (`resnet_cifar_main_horovod.py:218-221`).
  """

  def __init__(self, root_rank: int = 0):
    super().__init__()
    self.root_rank = root_rank
    self.broadcast_done = False

  def on_train_begin(self, logs=None):
    if self.broadcast_done:
      return
    broadcast_variables(self.model.store, self.root_rank)
    for ex in getattr(self.model, "_executors", {}).values():
      ex.params_updated()
    self.broadcast_done = True


class MetricAverageCallback(Callback):
  """Average the epoch-end metrics over all ranks (in place in `logs`)."""

  def on_epoch_end(self, epoch, logs=None):
    if not logs or size() <= 1:
      return
    keys = sorted(logs)
    dev = _c().device
    t = torch.tensor([float(logs[k]) for k in keys], dtype=torch.float64, device=dev)
    _c().all_reduce(t, average=True)
    for k, v in zip(keys, t.tolist()):
      logs[k] = np.float64(v)


class LearningRateScheduleCallback(Callback):
  """`lr = initial_lr * multiplier(epoch)` between `start_epoch` and `end_epoch`."""

  def __init__(self, multiplier, start_epoch=0, end_epoch=None, staircase=True,
               momentum_correction=True, steps_per_epoch=None, initial_lr=None):
    super().__init__()
    self.start_epoch = start_epoch
    self.end_epoch = end_epoch
    self.staircase = staircase
    self.momentum_correction = momentum_correction
    self.initial_lr = initial_lr
    self.steps_per_epoch = steps_per_epoch
    self.current_epoch = None
    self.multiplier = multiplier if callable(multiplier) else (lambda epoch: multiplier)

  def _in_range(self):
    return self.current_epoch >= self.start_epoch and (
        self.end_epoch is None or self.current_epoch < self.end_epoch)

  def on_train_begin(self, logs=None):
    if self.initial_lr is None:
      self.initial_lr = self.model.optimizer.current_lr()
    if not self.staircase and not self.steps_per_epoch:
      self.steps_per_epoch = self.params.get("steps")

  def on_epoch_begin(self, epoch, logs=None):
    self.current_epoch = epoch

  def _adjust(self, epoch):
    self.model.optimizer.learning_rate = float(self.initial_lr * self.multiplier(epoch))

  def on_batch_begin(self, batch, logs=None):
    if not self._in_range():
      return
    if self.staircase and batch == 0:
      self._adjust(self.current_epoch)
    elif not self.staircase:
      self._adjust(self.current_epoch + float(batch) / self.steps_per_epoch)

  def on_epoch_end(self, epoch, logs=None):
    if logs is not None:
      logs["lr"] = np.float64(self.model.optimizer.current_lr())


class LearningRateWarmupCallback(LearningRateScheduleCallback):
  """Gradual warm-up `lr = lr0 * (1/size) * (epoch * (size - 1) / warmup + 1)`.

  The multiplier ramps from `1/size` to 1 over the first `warmup_epochs`
  (fractional epochs, updated every batch) - "Accurate, Large Minibatch SGD".
  """

  def __init__(self, warmup_epochs=5, momentum_correction=True, steps_per_epoch=None,
               verbose=0, initial_lr=None):
    def multiplier(epoch):
      # Round-off guard as in Horovod: epoch is shifted by one step.
      epoch += 1.0 / self.steps_per_epoch
      return 1.0 / size() * (epoch * (size() - 1) / warmup_epochs + 1)
    super().__init__(multiplier, start_epoch=0, end_epoch=warmup_epochs, staircase=False,
                     momentum_correction=momentum_correction,
                     steps_per_epoch=steps_per_epoch, initial_lr=initial_lr)
    self.verbose = verbose
    self.warmup_epochs = warmup_epochs

  def on_epoch_end(self, epoch, logs=None):
    super().on_epoch_end(epoch, logs)
    if epoch == self.end_epoch - 1 and self.verbose > 0:
      print("\nEpoch %d: finished gradual learning rate warmup to %g."
            % (epoch + 1, self.model.optimizer.current_lr()))


callbacks = types.SimpleNamespace(
    BroadcastGlobalVariablesCallback=BroadcastGlobalVariablesCallback,
    MetricAverageCallback=MetricAverageCallback,
    LearningRateScheduleCallback=LearningRateScheduleCallback,
    LearningRateWarmupCallback=LearningRateWarmupCallback)
