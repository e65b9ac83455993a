"""Data planes of the parameter server: how a shard of parameters or gradients travels between a
worker and a PS rank.

The reference moves them as gRPC `RecvTensor` messages between TensorFlow servers
(`ps_server/log1.log:82-84`); here a transfer is a two-sided send / receive between two ranks of
the job, in one of two directions (PULL: PS -> worker, PUSH: worker -> PS), and both planes hand
back *work handles* (`wait()`, `is_completed()`):

* `TorchPlane` - `torch.distributed.isend / irecv` (gloo on the CPU and in rehearsals, RCCL through
  torch on request).  gloo moves one message with one progress thread per process group, so a
  large transfer is striped over several groups, half of them per direction.
* `RcclPlane` - the engine's own communicator (`ops/csrc/rccl_comm.cpp`): one communicator and one
  HIP stream *per direction*, so a receive that waits for its sender never sits in front of
  traffic going the other way, and a rank's pushes and pulls progress independently.  Transfers
  are stream-ordered device-to-device copies over xGMI (every worker has its own link to the PS:
  seven pushes can be in flight on seven links); a handle is an event on the direction's stream,
  and work that consumes received data can be ordered behind it on the device
  (`order_after`) without the host ever waiting.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

__all__ = ["PULL", "PUSH", "TorchPlane", "RcclPlane", "make_plane"]

PULL, PUSH = 0, 1              # directions: PS -> worker, worker -> PS


class _EventWork:
  """Completion of everything enqueued on a stream up to a point."""

  def __init__(self, event: "torch.cuda.Event"):
    self.event = event

  def wait(self):
    self.event.synchronize()

  def is_completed(self) -> bool:
    return self.event.query()


class TorchPlane:
  """`torch.distributed` point-to-point transfers, striped over `channels` process groups."""

  device_side = False            # received data is consumed after a host-side wait
  STRIPE_MIN = 1 << 17           # elements: smaller messages travel whole on the first channel

  def __init__(self, comm, channels: int = 1):
    self.comm = comm
    self.backend = comm.backend
    self._channels: List[List] = [[None], [None]]      # per direction: process groups
    if comm.world_size > 1 and channels > 1:
      import torch.distributed as dist  # pylint: disable=import-outside-toplevel
      groups = [dist.new_group(ranks=list(range(comm.world_size)), backend=comm.backend)
                for _ in range(channels)]
      self._channels = [groups[0::2], groups[1::2] or groups[0::2]]

  def staged(self, device) -> bool:
    """Device tensors are staged through pinned host buffers (any backend but RCCL: gloo's own
    handling of device tensors took seconds for the 102 MB of ResNet-50)."""
    return torch.device(device).type == "cuda" and self.backend != "nccl"

  def _stripes(self, t: torch.Tensor, direction: int):
    groups = self._channels[direction]
    n = t.numel()
    if len(groups) == 1 or n < self.STRIPE_MIN:
      return [(t, groups[0])]
    per = -(-n // len(groups))
    per = -(-per // 8) * 8
    return [(t[i * per:min(n, (i + 1) * per)], g) for i, g in enumerate(groups) if i * per < n]

  def isend(self, t: torch.Tensor, peer: int, direction: int) -> List:
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    return [dist.isend(c, peer, group=g) for c, g in self._stripes(t, direction)]

  def irecv(self, t: torch.Tensor, peer: int, direction: int) -> List:
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    return [dist.irecv(c, peer, group=g) for c, g in self._stripes(t, direction)]

  def describe(self) -> str:
    return "torch.distributed (%s), %d channel(s) per direction" % (
        self.backend, len(self._channels[0]))


class RcclPlane:
  """The engine's RCCL communicator, one per direction, each on a stream of its own."""

  device_side = True

  def __init__(self, comm, probe_against: Optional["torch.cuda.Stream"] = None):
    """COLLECTIVE: every rank of the job constructs its plane at the same point (the two
    communicators are created in the same order everywhere).  `probe_against`: a worker passes
    its compute stream, and gets transfer streams that were probed to run concurrently with it
    (HIP multiplexes streams onto a few hardware queues; two streams on one queue run strictly in
    submission order, and a send waiting for the PS would hold up the backward pass)."""
    from . import rccl  # pylint: disable=import-outside-toplevel
    from ..ops import native  # pylint: disable=import-outside-toplevel
    self.comm = comm
    self.backend = "nccl"
    self.device = comm.device
    self._comms = [rccl.new_native_comm(comm), rccl.new_native_comm(comm)]
    with torch.cuda.device(self.device):
      if probe_against is not None:
        hs = [native.ext().concurrent_stream(int(probe_against.cuda_stream), 0) for _ in range(2)]
        self._streams = [torch.cuda.ExternalStream(h, device=self.device) for h in hs]
      else:
        self._streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]

  def staged(self, device) -> bool:
    return False                 # device memory moves directly

  def stream(self, direction: int) -> "torch.cuda.Stream":
    return self._streams[direction]

  def order_after(self, direction: int, what):
    """Transfers issued from now on in `direction` start after `what` (an event or a stream)."""
    s = self._streams[direction]
    if isinstance(what, torch.cuda.Event):
      s.wait_event(what)
    else:
      s.wait_stream(what)

  def _done(self, direction: int) -> List:
    ev = torch.cuda.Event()
    ev.record(self._streams[direction])
    return [_EventWork(ev)]

  def isend(self, t: torch.Tensor, peer: int, direction: int) -> List:
    with torch.cuda.device(self.device):
      self._comms[direction].send(t, peer, stream=self._streams[direction].cuda_stream)
      return self._done(direction)

  def irecv(self, t: torch.Tensor, peer: int, direction: int) -> List:
    with torch.cuda.device(self.device):
      self._comms[direction].recv(t, peer, stream=self._streams[direction].cuda_stream)
      return self._done(direction)

  def exchange(self, sends, recvs, direction: int) -> List:
    """Several transfers of one direction as ONE group (they progress together): `sends` and
    `recvs` are lists of (tensor, peer).  A rank talking to itself (a one-rank rehearsal) must
    put the send and its matching receive into one group."""
    c = self._comms[direction]
    s = self._streams[direction].cuda_stream
    with torch.cuda.device(self.device):
      with c.group():
        for t, peer in recvs:
          c.recv(t, peer, stream=s)
        for t, peer in sends:
          c.send(t, peer, stream=s)
      return self._done(direction)

  def check(self):
    for c in self._comms:
      c.check()

  def describe(self) -> str:
    return "native RCCL communicators (one per direction, own streams)"


def make_plane(comm, requested: Optional[str], channels: Optional[int] = None,
               probe_against=None):
  """The data plane of a parameter-server job: the engine's RCCL communicators when the job runs
  on RCCL (`--comm native`, the default there), else `torch.distributed`."""
  from . import rccl  # pylint: disable=import-outside-toplevel
  kind = rccl.comm_choice(comm.backend, comm.device.type, requested)
  if kind == "native" and (comm.world_size > 1 or getattr(comm, "rehearse", False)):
    return RcclPlane(comm, probe_against)
  if channels is None:
    channels = int(os.environ.get("DTF_PS_CHANNELS", "1" if comm.backend == "nccl" else "4"))
  return TorchPlane(comm, channels)
