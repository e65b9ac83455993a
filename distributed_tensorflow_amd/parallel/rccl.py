"""The engine's own RCCL communicator (`ops/csrc/rccl_comm.cpp`) behind a small Python handle.

`torch.distributed` stays the default data plane (`parallel/comm.py`); this communicator is the
C++ alternative for the gradient collectives (`DTF_COMM=native`): the calls are issued from the
extension on a HIP stream handle, can be recorded as launch-plan entries (op kind `allreduce`)
and are therefore captured into the step's hipGraph, and the algorithm is chosen by message size -
one-shot `ncclAllReduce` for latency-bound buckets (all of ResNet-56 is 3.4 MB), reduce-scatter +
all-gather for large ones: xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), a
ring drives one link per direction while the direct exchange of 1/N shards uses all of them.

The 128-byte unique id travels through the rendezvous store of the process group that is already
up (the reference bootstraps its transports from `TF_CONFIG` the same way: gRPC server start-up,
`resnet_cifar_main_dist.py:38-50`).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from ..utils import logging

__all__ = ["NativeComm", "get_native_comm"]

_NATIVE: Optional["NativeComm"] = None
_SEQ = 0


def _rccl_library() -> str:
  """The RCCL build PyTorch itself loaded (one copy per process)."""
  path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
  return path if os.path.exists(path) else ""


class NativeComm:
  DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}

  def __init__(self, rank: int, world_size: int, store=None, device: Optional[torch.device] = None):
    from ..ops import native  # pylint: disable=import-outside-toplevel
    global _SEQ
    ext = native.ext()
    ext.RcclComm.load_library(_rccl_library())
    self.rank, self.world_size = int(rank), int(world_size)
    self.device = device or torch.device("cuda", torch.cuda.current_device())
    key = "dtf/rccl_unique_id/%d" % _SEQ
    _SEQ += 1
    with torch.cuda.device(self.device):
      if self.world_size == 1 or store is None:
        if self.world_size != 1:
          raise ValueError("a communicator of several ranks needs the rendezvous store")
        uid = ext.RcclComm.unique_id()
      elif self.rank == 0:
        uid = ext.RcclComm.unique_id()
        store.set(key, uid)
      else:
        uid = bytes(store.get(key))
      self._c = ext.RcclComm(uid, self.world_size, self.rank)
    min_mb = os.environ.get("DTF_ALLREDUCE_DIRECT_MIN_MB")
    if min_mb is not None:
      self._c.direct_min_bytes = int(float(min_mb) * (1 << 20))
    logging.info("native RCCL communicator: rank %d / %d, direct all-reduce from %d bytes",
                 self.rank, self.world_size, self._c.direct_min_bytes)

  @property
  def handle(self) -> int:
    """Address of the C++ object, for launch-plan entries (`allreduce` op)."""
    return self._c.handle

  @property
  def collectives(self) -> int:
    return self._c.collectives

  def _args(self, t: torch.Tensor, stream):
    if not t.is_cuda or not t.is_contiguous() or t.dtype not in self.DT:
      raise ValueError("native collectives take contiguous bf16 / fp16 / fp32 device tensors")
    s = int(torch.cuda.current_stream(t.device).cuda_stream) if stream is None else int(stream)
    return t.data_ptr(), t.numel(), self.DT[t.dtype], s

  def all_reduce(self, t: torch.Tensor, mode: str = "auto", stream=None):
    """In-place sum over the ranks on `stream` (default: torch's current stream).  mode:
    `oneshot` (ncclAllReduce), `direct` (reduce-scatter + all-gather), `auto` (by size)."""
    p, n, dt, s = self._args(t, stream)
    if mode == "oneshot":
      self._c.all_reduce(p, n, dt, s)
    elif mode == "direct" and n % self.world_size == 0:
      self._c.all_reduce_direct(p, n, dt, s)
    else:
      self._c.all_reduce_auto(p, n, dt, s)

  def broadcast(self, t: torch.Tensor, root: int = 0, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.broadcast(p, n, dt, root, s)

  def reduce(self, t: torch.Tensor, root: int = 0, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.reduce(p, n, dt, root, s)

  def plan_entry(self, t: torch.Tensor, mode: str = "auto") -> dict:
    """Descriptor of an `allreduce` launch-plan entry for `t` (captured with the plan)."""
    p, n, dt, _ = self._args(t, 0)
    return dict(comm=self.handle, buf=p, count=n, dt=dt,
                mode={"oneshot": 0, "direct": 1, "auto": 2}[mode])


def get_native_comm(comm) -> Optional[NativeComm]:
  """The process-wide native communicator over the ranks of `comm` (a `parallel.comm.Communicator`),
  created on first use; None when the job runs on the CPU."""
  global _NATIVE
  if _NATIVE is None:
    if comm.device.type != "cuda":
      return None
    store = comm.store() if (comm.world_size > 1 or comm.rehearse) else None
    _NATIVE = NativeComm(comm.rank, comm.world_size, store if comm.world_size > 1 else None,
                         comm.device)
  return _NATIVE
