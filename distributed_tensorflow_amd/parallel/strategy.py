"""Distribution strategies selectable with `--distribution_strategy`.

Semantics follow the reference's `distribution_utils.get_distribution_strategy`
(call site `resnet_imagenet_main.py:112-117`, documented in
`image_classification/README.md:89-99`), re-designed for one process per GPU:

| flag value              | reference                          | here                                              |
|-------------------------|------------------------------------|---------------------------------------------------|
| `off`                   | no strategy                        | `None`                                            |
| `one_device`, `default` with <= 1 GPU | `OneDeviceStrategy`  | `OneDeviceStrategy` (single rank)                 |
| `mirrored`, `default` with > 1 GPU    | `MirroredStrategy`, in-process NCCL | `MirroredStrategy`: one rank per GPU, bucketed RCCL all-reduce |
| `multi_worker_mirrored` | `MultiWorkerMirroredStrategy`      | same reducer, world = tasks x GPUs from `TF_CONFIG` |
| `parameter_server`      | `ParameterServerStrategy` (async)  | `ParameterServerStrategy` in `parameter_server.py` |

The global batch is split evenly over the replicas; each replica averages over
its shard, the all-reduce sums, the optimizer divides by the replica count - the
same effective update as Keras' `SUM_OVER_BATCH_SIZE` loss reduction under
`MirroredStrategy`.  BatchNorm statistics stay per replica (no SyncBN), as in
the reference.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch

from ..utils import logging
from .allreduce import GradientReducer
from .cluster import TaskInfo, resolve_task
from .comm import Communicator, get_communicator

__all__ = ["set_up_synthetic_data", "undo_set_up_synthetic_data",
           "Strategy", "OneDeviceStrategy", "MirroredStrategy",
           "MultiWorkerMirroredStrategy", "get_distribution_strategy",
           "get_strategy_scope", "strategy_needs_launch"]


class Strategy:
  """Base strategy = one replica on one device."""

  name = "one_device"
  synchronous = True

  def __init__(self, device: Optional[torch.device] = None):
    if device is None:
      device = torch.device("cuda", torch.cuda.current_device()) \
          if torch.cuda.is_available() else torch.device("cpu")
    self.device = torch.device(device)
    self.comm: Optional[Communicator] = None
    self.task = TaskInfo()
    self.all_reduce_alg = None
    self.num_packs = 1
    self.bucket_bytes = None

  # -- topology --------------------------------------------------------------------
  @property
  def num_replicas_in_sync(self) -> int:
    return 1

  @property
  def rank(self) -> int:
    return 0

  @property
  def is_chief(self) -> bool:
    return True

  @property
  def is_ps_task(self) -> bool:
    return False

  def per_replica_batch_size(self, global_batch: int) -> int:
    n = self.num_replicas_in_sync
    if global_batch % n:
      raise ValueError("Batch size must be divisible by the number of replicas: "
                       "%d vs %d" % (global_batch, n))
    return global_batch // n

  def scope(self):
    return contextlib.nullcontext(self)

  # -- hooks used by Model ------------------------------------------------------------
  def make_reducer(self, store, buckets) -> Optional[GradientReducer]:
    return None

  def broadcast_variables(self, store, root: int = 0):
    """Makes every replica start from rank `root`'s variables."""

  def reduce_scalars(self, t: torch.Tensor) -> torch.Tensor:
    """Sums a small tensor of metric accumulators over the replicas."""
    return t

  def average_state(self, state: torch.Tensor) -> torch.Tensor:
    """Mean over the replicas of the non-trainable state (BatchNorm moving statistics), in
    place.  Every replica normalises with its own batch statistics (no SyncBN, as in the
    reference), so the moving averages drift apart; Keras declares them ON_READ / MEAN under
    `MirroredStrategy`: reading them for a checkpoint or an export yields the replica mean."""
    return state

  def barrier(self):
    pass

  def shutdown(self):
    pass


class OneDeviceStrategy(Strategy):
  name = "one_device"


class MirroredStrategy(Strategy):
  """Synchronous data parallelism: one rank per GPU, fused-bucket RCCL all-reduce."""

  name = "mirrored"

  def __init__(self, task: Optional[TaskInfo] = None, use_gpu: bool = True,
               all_reduce_alg: Optional[str] = None, num_packs: int = 1,
               bucket_bytes: Optional[int] = None, comm_kind: Optional[str] = None):
    task = task or resolve_task()
    comm = get_communicator(task, use_gpu=use_gpu)
    super().__init__(comm.device)
    self.comm = comm
    self.task = task
    self.comm_kind = comm_kind           # `--comm`: native / torch / None (auto)
    self.all_reduce_alg = all_reduce_alg
    self.num_packs = num_packs
    self.bucket_bytes = bucket_bytes

  @property
  def num_replicas_in_sync(self) -> int:
    return self.comm.world_size

  @property
  def rank(self) -> int:
    return self.comm.rank

  @property
  def is_chief(self) -> bool:
    return self.comm.rank == 0

  def make_reducer(self, store, buckets):
    if self.comm.world_size <= 1 and not self.comm.rehearse:
      return None
    return GradientReducer(self.comm, store, buckets, self.all_reduce_alg,
                           comm_kind=self.comm_kind)

  def broadcast_variables(self, store, root: int = 0):
    if self.comm.world_size <= 1 and not self.comm.rehearse:
      return
    self.comm.broadcast(store.params, root)
    self.comm.broadcast(store.state, root)
    for name in sorted(store.slots):
      self.comm.broadcast(store.slots[name], root)
    store.version += 1

  def reduce_scalars(self, t):
    if self.comm.world_size > 1 or self.comm.rehearse:
      self.comm.all_reduce(t)
    return t

  def average_state(self, state):
    if state.numel() and (self.comm.world_size > 1 or self.comm.rehearse):
      self.comm.all_reduce(state, average=True)
    return state

  def barrier(self):
    self.comm.barrier()


class MultiWorkerMirroredStrategy(MirroredStrategy):
  """Same engine as `MirroredStrategy`; ranks span the `TF_CONFIG` workers."""

  name = "multi_worker_mirrored"


def strategy_needs_launch(distribution_strategy: str, num_gpus: int) -> int:
  """How many local ranks a *directly started* driver has to fan out into.

  The reference runs N GPUs inside one process; here a driver invoked without a
  launcher environment re-launches itself once per GPU (`launch.py`).  Returns 0
  when no fan-out is needed.
  """
  import os  # pylint: disable=import-outside-toplevel
  if "RANK" in os.environ:       # already one of the ranks
    return 0
  ds = (distribution_strategy or "default").lower()
  if ds in ("mirrored", "multi_worker_mirrored", "default") and num_gpus > 1:
    return num_gpus
  return 0


def get_distribution_strategy(distribution_strategy="default", num_gpus=0, num_workers=1,
                              all_reduce_alg=None, num_packs=1, bucket_bytes=None, comm_kind=None):
  """Returns a strategy object, or None for `off`.

  Args mirror the reference helper.  `num_gpus` is GPUs *per worker*; 0 runs on the
  CPU (with the `gloo` backend when more than one rank exists).
  """
  if num_gpus < 0:
    raise ValueError("`num_gpus` can not be negative.")
  ds = (distribution_strategy or "default").lower()
  use_gpu = num_gpus > 0

  if ds == "off":
    if num_gpus > 1:
      raise ValueError("When {} GPUs are specified, distribution_strategy flag cannot be "
                       "set to 'off'.".format(num_gpus))
    return None

  if ds == "multi_worker_mirrored":
    task = resolve_task(gpus_per_task=max(num_gpus, 1))
    return MultiWorkerMirroredStrategy(task, use_gpu, all_reduce_alg, num_packs, bucket_bytes,
                                       comm_kind)

  if ds == "one_device" or (ds == "default" and num_gpus <= 1):
    if num_gpus > 1:
      raise ValueError("`OneDeviceStrategy` can not be used for more than one device.")
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu and \
        torch.cuda.is_available() else torch.device("cpu")
    return OneDeviceStrategy(dev)

  if ds in ("mirrored", "default"):
    task = resolve_task(gpus_per_task=max(num_gpus, 1))
    if all_reduce_alg not in (None, "nccl", "hierarchical_copy", "ring", "mesh"):
      raise ValueError("When used with `mirrored`, valid values for all_reduce_alg are "
                       "['nccl', 'hierarchical_copy', 'ring', 'mesh']. Supplied value: {}"
                       .format(all_reduce_alg))
    return MirroredStrategy(task, use_gpu, all_reduce_alg, num_packs, bucket_bytes, comm_kind)

  if ds == "parameter_server":
    from .parameter_server import ParameterServerStrategy  # pylint: disable=import-outside-toplevel
    task = resolve_task(gpus_per_task=1)
    return ParameterServerStrategy(task, use_gpu, comm_kind=comm_kind)

  raise ValueError("Unrecognized Distribution Strategy: %r" % distribution_strategy)


def set_up_synthetic_data():
  """`distribution_utils.set_up_synthetic_data`: keep the synthetic batch on the device."""
  from ..data import synthetic  # pylint: disable=import-outside-toplevel
  synthetic.set_up_synthetic_data()


def undo_set_up_synthetic_data():
  from ..data import synthetic  # pylint: disable=import-outside-toplevel
  synthetic.undo_set_up_synthetic_data()


def get_strategy_scope(strategy):
  return strategy.scope() if strategy else contextlib.nullcontext()
