"""Asynchronous parameter-server training over point-to-point transfers.

What the reference does (`ps_server/resnet_imagenet_main_dist_ps_{0..15}.py`,
`README.md:165-194`): one `ps` task owns every variable, 15 single-GPU `worker`
tasks each pull the parameters, compute a gradient on their own batch and push it
back over gRPC; the PS applies SGD-momentum on arrival, with no barrier between
workers (stale gradients), and never exits on its own (`ps_server/kill.sh`).

Here the PS ranks hold contiguous shards of the flat parameter / momentum
buffers in HBM and the data plane is `send/recv` on the RCCL (or gloo) process
group.  Two-sided transfers need both ends to agree on who talks next, so the
control plane is a FIFO mailbox per PS rank kept in the rendezvous TCP store:

    worker:  i = store.add("ps<k>/tail", 1); store.set("ps<k>/req/<i-1>", "<op>:<rank>:<args>")
             then send()/recv() the payload with PS k
    PS k:    waits for "ps<k>/req/<head>", serves it, head += 1

which gives any-source, first-come-first-served service without deadlock: a PS
only ever posts a transfer that the requesting worker is about to post too.

Ops: `pull` (PS -> worker: parameter shard, PS 0 also BatchNorm moving
statistics), `push:<lr>:<divisor>` (worker -> PS: gradient shard, to PS 0 also
moving statistics; the PS runs the fused SGD-momentum+L2 update on its shard
with the learning rate the worker's schedule computed), `done` (worker is
finished; a PS returns once every worker said so - a clean shutdown instead of
`pkill`).
"""
from __future__ import annotations

import datetime
import os
import time
from typing import List, Optional, Tuple

import torch

from ..utils import logging
from .cluster import TaskInfo, resolve_task
from .comm import Communicator, get_communicator
from .strategy import Strategy

__all__ = ["ParameterServerStrategy", "shard_ranges"]


def shard_ranges(n: int, parts: int, align: int = 4) -> List[Tuple[int, int]]:
  """Splits [0, n) into `parts` contiguous, `align`-aligned ranges."""
  per = -(-n // parts)
  per = -(-per // align) * align
  out = []
  for k in range(parts):
    a = min(k * per, n)
    b = min(a + per, n)
    out.append((a, b))
  return out


class ParameterServerStrategy(Strategy):
  name = "parameter_server"
  synchronous = False

  def __init__(self, task: Optional[TaskInfo] = None, use_gpu: bool = True,
               poll_timeout_s: float = 1800.0):
    task = task or resolve_task(gpus_per_task=1)
    if task.num_ps < 1:
      raise ValueError("parameter_server strategy needs at least one 'ps' task in the "
                       "cluster (TF_CONFIG) or DTF_NUM_PS >= 1 from the launcher")
    if task.num_workers < 1:
      raise ValueError("parameter_server strategy needs at least one worker")
    if task.job == "ps" and os.environ.get("DTF_PS_DEVICE", "").lower() == "cpu":
      # shard holder in host memory (needs a backend that moves CPU tensors, i.e. gloo): for
      # rehearsals on a machine with fewer GPUs than ranks, and for models that should not
      # spend a GPU on holding parameters
      use_gpu = False
    comm = get_communicator(task, use_gpu=use_gpu)
    super().__init__(comm.device)
    self.comm = comm
    self.task = task
    self.num_ps = task.num_ps
    self.num_workers = task.num_workers
    self.poll_timeout_s = poll_timeout_s
    self._store = comm.store() if comm.world_size > 1 else None
    self._done_sent = False
    self.updates_applied = 0
    self._host = {}

  # -- topology --------------------------------------------------------------------
  @property
  def num_replicas_in_sync(self) -> int:
    return 1          # asynchronous: every worker steps on its own

  @property
  def rank(self) -> int:
    return self.comm.rank

  @property
  def is_ps_task(self) -> bool:
    return self.task.job == "ps"

  @property
  def is_chief(self) -> bool:
    return (not self.is_ps_task) and self.task.worker_rank == 0

  @property
  def worker_index(self) -> int:
    return self.task.worker_rank

  def per_replica_batch_size(self, global_batch: int) -> int:
    return global_batch      # `--batch_size` is per worker process (ps_server/run.sh:5)

  # -- mailbox ---------------------------------------------------------------------
  def _post(self, ps: int, msg: str):
    i = self._store.add("ps%d/tail" % ps, 1) - 1
    self._store.set("ps%d/req/%d" % (ps, i), msg)

  def _next_request(self, ps: int, head: int) -> str:
    key = "ps%d/req/%d" % (ps, head)
    deadline = time.time() + self.poll_timeout_s
    while True:
      try:
        self._store.wait([key], datetime.timedelta(seconds=2))
        break
      except Exception:  # timeout: keep polling until the global deadline
        if time.time() > deadline:
          raise TimeoutError("parameter server %d saw no request for %.0f s"
                             % (ps, self.poll_timeout_s)) from None
    msg = self._store.get(key).decode()
    try:
      self._store.delete_key(key)
    except Exception:  # pragma: no cover - stores without delete
      pass
    return msg

  # -- transfers ---------------------------------------------------------------------
  # RCCL moves device memory directly.  Any other backend (gloo: parameter server in host
  # memory like the reference's `/job:ps/.../CPU:0`, or a rehearsal on fewer GPUs than ranks)
  # gets device tensors staged through pinned host buffers here: gloo's own handling of device
  # tensors took seconds for the 102 MB of ResNet-50 on the test machine.
  def _staging(self, t: torch.Tensor) -> Optional[torch.Tensor]:
    if not t.is_cuda or self.comm.backend == "nccl":
      return None
    key = (t.numel(), t.dtype)
    buf = self._host.get(key)
    if buf is None:
      buf = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
      self._host[key] = buf
    return buf

  def _send(self, t: torch.Tensor, dst: int):
    h = self._staging(t)
    if h is None:
      self.comm.send(t, dst)
    else:
      h.copy_(t.reshape(-1))             # synchronous device-to-host copy
      self.comm.send(h, dst)

  def _recv(self, t: torch.Tensor, src: int):
    h = self._staging(t)
    if h is None:
      self.comm.recv(t, src)
    else:
      self.comm.recv(h, src)
      t.reshape(-1).copy_(h, non_blocking=False)

  # -- worker side -------------------------------------------------------------------
  def _shards(self, store):
    return shard_ranges(store.num_elements, self.num_ps)

  def pull(self, store, half: bool = False):
    """Fetch the current parameters (and moving statistics) from the PS ranks.

    `half`: the worker computes in a 16-bit format, so it asks for bfloat16 parameters - half
    the bytes per pull, and exactly the values its executor would round the fp32 ones to.  The
    master copy stays fp32 on the PS (updates are applied there)."""
    for k, (a, b) in enumerate(self._shards(store)):
      self._post(k, "pull:%d:%s" % (self.comm.rank, "bf16" if half else "fp32"))
    for k, (a, b) in enumerate(self._shards(store)):
      if b > a and half:
        buf = self._half_buf(b - a, store.device)
        self._recv(buf, k)
        store.params[a:b].copy_(buf)
      elif b > a:
        self._recv(store.params[a:b], k)
      if k == 0 and store.num_state_elements:
        self._recv(store.state, 0)
    store.version += 1

  def _half_buf(self, n: int, device) -> torch.Tensor:
    key = ("half", n, str(device))
    if key not in self._host:
      self._host[key] = torch.empty(n, dtype=torch.bfloat16, device=device)
    return self._host[key]

  def push(self, store, lr: float, divisor: float = 1.0):
    """Send this worker's gradients; the PS applies them on arrival."""
    for k, (a, b) in enumerate(self._shards(store)):
      self._post(k, "push:%d:%r:%r" % (self.comm.rank, float(lr), float(divisor)))
    for k, (a, b) in enumerate(self._shards(store)):
      if b > a:
        self._send(store.grads[a:b], k)
      if k == 0 and store.num_state_elements:
        self._send(store.state, 0)

  def shutdown(self):
    if self.is_ps_task or self._done_sent or self._store is None:
      return
    for k in range(self.num_ps):
      self._post(k, "done:%d" % self.comm.rank)
    self._done_sent = True

  # -- PS side ---------------------------------------------------------------------
  def serve(self, model) -> int:
    """Request loop of a PS rank. Returns the number of updates applied."""
    store = model.store
    opt = model.optimizer
    k = self.task.task_index
    a, b = self._shards(store)[k]
    n = b - a
    recv_buf = torch.zeros(max(n, 1), dtype=torch.float32, device=store.device)
    shard = _ShardView(store, a, b)
    head = 0
    active = self.num_workers
    logging.info("parameter server %d/%d serving shard [%d, %d) for %d workers",
                 k, self.num_ps, a, b, active)
    t0 = time.time()
    while active > 0:
      msg = self._next_request(k, head)
      head += 1
      parts = msg.split(":")
      op, src = parts[0], int(parts[1])
      if op == "pull":
        if n and len(parts) > 2 and parts[2] == "bf16":
          buf = self._half_buf(n, store.device)
          buf.copy_(store.params[a:b])
          self._send(buf, src)
        elif n:
          self._send(store.params[a:b], src)
        if k == 0 and store.num_state_elements:
          self._send(store.state, src)
      elif op == "push":
        lr, divisor = float(parts[2]), float(parts[3])
        if n:
          self._recv(recv_buf[:n], src)
        if k == 0 and store.num_state_elements:
          self._recv(store.state, src)
        if n:
          shard.apply_sgd(recv_buf[:n], lr, getattr(opt, "momentum", 0.0), divisor)
        self.updates_applied += 1
      elif op == "done":
        active -= 1
      else:
        raise RuntimeError("parameter server: bad request %r" % msg)
    if store.device.type == "cuda":
      torch.cuda.synchronize(store.device)
    logging.info("parameter server %d: %d updates applied in %.1f s; all workers done",
                 k, self.updates_applied, time.time() - t0)
    return self.updates_applied

  def barrier(self):
    pass   # asynchronous by design: no inter-worker barrier


class _ShardView:
  """The slice of the flat buffers one PS rank is authoritative for."""

  def __init__(self, store, a: int, b: int):
    self.store = store
    self.a, self.b = a, b

  def apply_sgd(self, grads: torch.Tensor, lr: float, momentum: float, divisor: float):
    from ..ops import fused  # pylint: disable=import-outside-toplevel
    st = self.store
    a, b = self.a, self.b
    fused.sgd_momentum_l2(st.params[a:b], grads, st.slot("momentum")[a:b],
                          st.l2_vector()[a:b], lr, momentum, 1.0 / divisor)
    st.version += 1
