"""Process-group bootstrap and the collective / point-to-point primitives.

One process per GPU; the data plane is `torch.distributed` with the `nccl`
backend, which on ROCm *is* RCCL (xGMI inside a node), and `gloo` on CPU-only
hosts (used by the multi-process tests).  This replaces the four transports of
the reference (in-process NCCL for `MirroredStrategy`, gRPC ring / NCCL
`CollectiveOps` for `MultiWorkerMirroredStrategy`, gRPC `RecvTensor` for the
parameter server, MPI+NCCL for Horovod: `README.md:61-77`, `ps_server/log1.log:82-84`).

Rendezvous uses a TCP store on the first task's address, taken from the
launcher environment or from the `TF_CONFIG` cluster spec, which plays the part
of the reference's gRPC server start-up / `horovodrun` MPI wire-up.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ..utils import logging
from .cluster import TaskInfo, resolve_task

__all__ = ["Communicator", "get_communicator", "shutdown"]

_COMM: Optional["Communicator"] = None


class Communicator:
  """Thin, explicit wrapper around the default process group."""

  def __init__(self, task: TaskInfo, device: torch.device, backend: str,
               owns_group: bool):
    self.task = task
    self.device = device
    self.backend = backend
    self.rank = task.rank
    self.world_size = task.world_size
    self._owns_group = owns_group
    self._groups = {}
    # Rehearsal: a one-rank process group whose collectives are really issued (see
    # `get_communicator`), so the stream / event choreography of the multi-GPU path can be
    # exercised and timed on a single GPU.
    self.rehearse = False

  def _single(self) -> bool:
    return self.world_size == 1 and not self.rehearse

  # -- collectives -----------------------------------------------------------------
  def all_reduce(self, t: torch.Tensor, average: bool = False, async_op: bool = False,
                 group=None):
    if self._single():
      return None
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op, group=group)
    if average:
      if async_op:
        raise ValueError("average with async_op: divide after wait()")
      t.div_(dist.get_world_size(group))
    return w

  def reduce(self, t: torch.Tensor, dst: int = 0, async_op: bool = False, group=None):
    if self._single():
      return None
    return dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op, group=group)

  def broadcast(self, t: torch.Tensor, root: int = 0, async_op: bool = False, group=None):
    if self._single():
      return None
    return dist.broadcast(t, src=root, async_op=async_op, group=group)

  def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op=False, group=None):
    return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, async_op=async_op,
                                      group=group)

  def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op=False, group=None):
    return dist.all_gather_into_tensor(out, inp, async_op=async_op, group=group)

  def supports_reduce_scatter(self) -> bool:
    return self.backend == "nccl"

  def send(self, t: torch.Tensor, dst: int):
    dist.send(t, dst)

  def recv(self, t: torch.Tensor, src: int):
    dist.recv(t, src)

  def isend(self, t, dst):
    return dist.isend(t, dst)

  def irecv(self, t, src):
    return dist.irecv(t, src)

  def barrier(self, group=None):
    if not self._single():
      if self.backend == "nccl":
        dist.barrier(group=group, device_ids=[self.device.index])
      else:
        dist.barrier(group=group)

  def new_group(self, ranks: List[int]):
    """Sub-group (cached); every rank of the job must call this with the same list."""
    key = tuple(ranks)
    if key not in self._groups:
      self._groups[key] = dist.new_group(ranks=list(ranks))
    return self._groups[key]

  def store(self):
    """The rendezvous key-value store (control plane for the parameter server)."""
    from torch.distributed import distributed_c10d as c10d  # pylint: disable=import-outside-toplevel
    return c10d._get_default_store()  # pylint: disable=protected-access

  def shutdown(self):
    if self._owns_group and dist.is_initialized():
      try:
        dist.destroy_process_group()
      except Exception as e:  # pragma: no cover
        logging.warning("destroy_process_group: %s", e)


def _pick_device(local_rank: int, use_gpu: bool) -> torch.device:
  if use_gpu and torch.cuda.is_available():
    n = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(n, 1))
    torch.cuda.set_device(dev)
    return dev
  return torch.device("cpu")


def get_communicator(task: Optional[TaskInfo] = None, use_gpu: bool = True,
                     backend: Optional[str] = None, timeout_s: int = 1800) -> Communicator:
  """Initialises (once) the process group described by `task` and returns the wrapper.

  A group that already exists (e.g. created by `torchrun`-style launch code or by a
  test harness) is adopted rather than re-created.
  """
  global _COMM
  if _COMM is not None:
    return _COMM
  task = task or resolve_task()
  device = _pick_device(task.local_rank, use_gpu)
  # DTF_BACKEND overrides the choice (e.g. gloo to run several ranks on one GPU in tests)
  backend = backend or os.environ.get("DTF_BACKEND") or (
      "nccl" if device.type == "cuda" else "gloo")
  owns = False
  if task.world_size > 1 and not dist.is_initialized():
    init_method = "tcp://%s:%d" % (task.master_addr, task.master_port)
    logging.info("init_process_group backend=%s rank=%d/%d via %s",
                 backend, task.rank, task.world_size, init_method)
    kwargs = {}
    if backend == "nccl":
      kwargs["device_id"] = device
    dist.init_process_group(backend, init_method=init_method, rank=task.rank,
                            world_size=task.world_size,
                            timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    owns = True
  elif dist.is_initialized():
    backend = dist.get_backend()
  rehearse = False
  if task.world_size == 1 and os.environ.get("DTF_REHEARSE_COLLECTIVES") == "1" \
      and not dist.is_initialized():
    # one-rank group over a private file store: every collective is issued for real
    import tempfile  # pylint: disable=import-outside-toplevel
    path = os.path.join(tempfile.mkdtemp(prefix="dtf_store_"), "store")
    kwargs = {"device_id": device} if backend == "nccl" else {}
    dist.init_process_group(backend, store=dist.FileStore(path, 1), rank=0, world_size=1,
                            timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    owns = rehearse = True
  _COMM = Communicator(task, device, backend, owns)
  _COMM.rehearse = rehearse
  return _COMM


def shutdown():
  global _COMM
  if _COMM is not None:
    _COMM.shutdown()
    _COMM = None
