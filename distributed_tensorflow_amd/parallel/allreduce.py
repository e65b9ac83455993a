"""Bucketed gradient all-reduce, overlapped with the backward pass.

The executor calls `on_bucket_ready(bucket)` as soon as every gradient of a
fusion bucket has been *enqueued* on the compute stream (buckets complete in
reverse layer order).  Each call launches one asynchronous collective on the
bucket's slice of the flat gradient buffer; with the RCCL backend the collective
runs on a dedicated communication stream after an event recorded on the compute
stream, so it overlaps the remaining backward kernels.  `finish()` makes the compute
stream wait for all outstanding collectives before the optimizer runs.

Algorithms (`--all_reduce_alg`, reference `resnet_imagenet_main.py:116-117`):

* `nccl` (default): one `ncclAllReduce` per bucket; RCCL picks ring/tree/direct.
* `ring`: explicit reduce-scatter + all-gather on the bucket (each GPU owns 1/N
  of the bucket between the two phases).
* `hierarchical_copy`: reduce to rank 0, then broadcast - the single-root
  counterpart of TF's `HierarchicalCopyAllReduce`.
* `mesh`: all-link exchange for a full mesh of point-to-point links (xGMI): every rank sends
  peer j its copy of shard j and receives j's copy of its own shard - all peers at once -, adds
  them up, then the reduced shards are exchanged the same way.  Each link carries 2/N of the
  bucket; a ring pushes 2(N-1)/N of it through one link per direction.

Data plane: the engine's own RCCL communicator (`parallel/rccl.py`, `--comm native`) whenever
the job runs on RCCL, `torch.distributed` otherwise (gloo on the CPU) or on request
(`--comm torch`).

The reduction is a SUM of per-replica mean gradients; the division by the
number of replicas is folded into the fused optimizer (`divisor`).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from ..engine.params import Bucket, ParamStore
from .comm import Communicator

__all__ = ["GradientReducer", "ALL_REDUCE_ALGS"]

ALL_REDUCE_ALGS = (None, "nccl", "rccl", "ring", "hierarchical_copy", "mesh")


def _priority_range(device):
  """(least urgent, most urgent) HIP stream priorities of `device`."""
  with torch.cuda.device(device):
    try:
      return torch.cuda.Stream.priority_range()
    except Exception:  # pylint: disable=broad-except
      return 0, -1


class GradientReducer:
  def __init__(self, comm: Communicator, store: ParamStore, buckets: List[Bucket],
               alg: Optional[str] = None, group=None, group_size: Optional[int] = None,
               comm_kind: Optional[str] = None):
    alg = alg.lower() if isinstance(alg, str) else alg
    if alg not in ALL_REDUCE_ALGS:
      raise ValueError("unknown all_reduce_alg %r (choose from nccl, ring, "
                       "hierarchical_copy, mesh)" % (alg,))
    self.comm_kind = comm_kind     # "native" / "torch" / None: decided at first use
    self.comm = comm
    self.store = store
    self.buckets = buckets
    self.alg = alg or "nccl"
    self.group = group
    self.n = group_size if group_size is not None else comm.world_size
    self._pending = []
    self._stream = None            # None: not decided yet; False: torch.distributed's own stream
    self._stream_pending = False
    self._side = None              # the executor's side lane, when it orders its results here
    self._executor = None          # set by orders_side_lane()
    self._native = None            # the engine's own RCCL communicator (DTF_COMM=native)
    self._in_plan = set()          # executors whose backward plan ends with the collective
    self._skip = False
    self.bytes_reduced = 0
    self.collectives_launched = 0

  @property
  def divisor(self) -> float:
    """What the summed gradients must be divided by to become the global mean."""
    return float(self.n)

  def _slice(self, b: Bucket) -> torch.Tensor:
    return self.store.grads[b.start:b.end]

  def _own_stream(self, device) -> Optional["torch.cuda.Stream"]:
    """The communication stream of the RCCL path: a stream probed (by the native extension) to
    execute concurrently with the compute stream and the executor's side lane, i.e. sitting on
    a hardware queue of its own.  A collective kernel that shared the compute stream's queue
    would run in submission order with the backward kernels and hold them up while it waits
    for its peers.  `DTF_COMM_STREAM=torch` leaves the choice to torch.distributed (its
    internal stream, `async_op=True`)."""
    if self._stream is None:
      self._stream = False
      # Small-message policy: a model whose whole gradient is one bucket (ResNet-56: 3.4 MB) has
      # nothing left to overlap when the bucket completes - the collective is issued from the
      # compute stream's own position (torch.distributed's stream waits for it and the optimizer
      # waits for the collective), which saves the two extra cross-stream hops of a private
      # communication stream.  DTF_COMM_STREAM=own forces the private stream.
      single = len(self.buckets) <= 1 and os.environ.get("DTF_COMM_STREAM") != "own"
      if (self.comm.backend == "nccl" and device.type == "cuda" and not single
          and os.environ.get("DTF_COMM_STREAM", "own") != "torch"):
        from ..ops import native  # pylint: disable=import-outside-toplevel
        if native.available():
          _lo, hi = _priority_range(device)
          # Normal priority: a high-priority communication stream was seen to stall the first
          # hipGraph replay that follows it (one-rank rehearsal on an MI355X); opt in with
          # DTF_COMM_PRIORITY=high for eager-launch experiments.
          prio = hi if os.environ.get("DTF_COMM_PRIORITY", "normal") == "high" else 0
          cur = torch.cuda.current_stream(device)
          h = native.ext().concurrent_stream(int(cur.cuda_stream), prio)
          self._stream = torch.cuda.ExternalStream(h, device=device)
    return self._stream or None

  def orders_side_lane(self, executor) -> bool:
    """True when every bucket's collective is issued on this reducer's own stream, which then
    also waits for the executor's side lane (weight gradients): the executor may skip the
    main-lane joins at bucket boundaries (`HipExecutor.train_step(soft_joins=True)`)."""
    if self.n <= 1 and not getattr(self.comm, "rehearse", False):
      return False
    if not hasattr(executor, "side_stream") or self.store.grads.device.type != "cuda":
      return False
    if self._own_stream(self.store.grads.device) is None:
      return False
    self._executor = executor
    return True

  def _native_comm(self):
    """The engine's own RCCL communicator: the data plane of GPU jobs unless `--comm torch` /
    `DTF_COMM=torch` asks for `torch.distributed` (`rccl.comm_choice`).  A sub-group reducer
    keeps `torch.distributed` (the native communicator spans the whole job)."""
    if self._native is None:
      self._native = False
      from . import rccl  # pylint: disable=import-outside-toplevel
      kind = rccl.comm_choice(self.comm.backend, self.store.grads.device.type, self.comm_kind)
      if kind == "native" and self.group is None:
        try:
          self._native = rccl.get_native_comm(self.comm) or False
        except Exception as e:  # pylint: disable=broad-except
          if "native" in ((self.comm_kind or "").lower(), os.environ.get("DTF_COMM", "").lower()):
            raise
          from ..utils import logging  # pylint: disable=import-outside-toplevel
          logging.warning("native RCCL communicator unavailable (%s): gradient collectives go "
                          "through torch.distributed", e)
      self.comm_kind = "native" if self._native else "torch"
    return self._native or None

  def describe(self) -> str:
    """One line for the start-up log: data plane, algorithm, buckets, stream."""
    nat = self._native_comm()
    dev = self.store.grads.device
    stream = "n/a"
    if dev.type == "cuda":
      cs = self._own_stream(dev)
      stream = "own probed stream" if cs is not None else "compute stream position (single bucket)" \
          if len(self.buckets) <= 1 else "torch.distributed's stream"
    sizes = [(b.end - b.start) * 4 / 2 ** 20 for b in self.buckets]
    return ("%s, alg %s, %d buckets (%s MB), collectives on: %s" % (
        "native RCCL communicator" if nat is not None else "torch.distributed (%s)" % self.comm.backend,
        self.alg, len(self.buckets), ", ".join("%.1f" % s for s in sizes), stream))

  def attach_to_plan(self, executor) -> bool:
    """Small-message policy, native communicator: the single bucket's all-reduce becomes the last
    entry of the executor's backward plan (captured with it); `on_bucket_ready` / `finish` then
    have nothing left to do for that executor.  False when this reducer or executor cannot."""
    if executor in self._in_plan:
      return True
    if (len(self.buckets) != 1 or not hasattr(executor, "attach_collective")
        or self.alg == "hierarchical_copy"
        or (self.n <= 1 and not getattr(self.comm, "rehearse", False))
        or os.environ.get("DTF_COMM_IN_PLAN", "1") == "0"):
      return False
    nat = self._native_comm()
    if nat is None:
      return False
    g = self._slice(self.buckets[0])
    if executor.attach_collective(nat.plan_entry(
        g, {"ring": "direct", "mesh": "mesh"}.get(self.alg, "auto"))):
      self._in_plan.add(executor)
      return True
    return False

  def _issue(self, g: torch.Tensor, async_op: bool):
    """The bucket's collective(s); returns the last work handle (None for blocking calls)."""
    c = self.comm
    nat = self._native_comm() if g.is_cuda else None
    if nat is not None:
      # stream-ordered on torch's current stream (the reducer's communication stream, or the
      # compute stream for a single bucket): nothing to wait for on the host
      if self.alg == "hierarchical_copy":
        nat.reduce(g, 0)
        nat.broadcast(g, 0)
        self.collectives_launched += 2
      else:
        before = nat.collectives
        nat.all_reduce(g, {"ring": "direct", "mesh": "mesh"}.get(self.alg, "auto"))
        self.collectives_launched += nat.collectives - before      # direct / mesh: two phases
      return None
    if self.alg == "mesh" and g.numel() % self.n == 0 and self.n > 1:
      self.collectives_launched += 2
      return self._mesh(g)
    if self.alg == "ring" and c.supports_reduce_scatter() and g.numel() % self.n == 0:
      shard = g.numel() // self.n
      r = c.rank if self.group is None else torch.distributed.get_rank(self.group)
      mine = g[r * shard:(r + 1) * shard]
      w1 = c.reduce_scatter(mine, g, async_op=async_op, group=self.group)
      if w1 is not None:
        w1.wait()   # stream-ordered on the comm stream; no host block with RCCL
      w2 = c.all_gather(g, mine, async_op=async_op, group=self.group)
      self.collectives_launched += 2
      return w2
    if self.alg == "hierarchical_copy":
      w1 = c.reduce(g, dst=0, async_op=async_op, group=self.group)
      if w1 is not None:
        w1.wait()
      w2 = c.broadcast(g, root=0, async_op=async_op, group=self.group)
      self.collectives_launched += 2
      return w2
    self.collectives_launched += 1
    return c.all_reduce(g, async_op=async_op, group=self.group)

  def on_bucket_ready(self, b: Bucket):
    if self.n <= 1 and not getattr(self.comm, "rehearse", False):
      return
    if self._skip:                 # the running executor's plan contains the collective
      self.bytes_reduced += (b.end - b.start) * 4
      return
    g = self._slice(b)
    c = self.comm
    self.bytes_reduced += g.numel() * 4
    if g.is_cuda and c.backend != "nccl" and os.environ.get("DTF_GLOO_DRAIN", "1") != "0":
      # gloo stages device tensors through the host on streams of its own; a rehearsal of
      # several ranks on one GPU once stalled in this path.  The stall no longer reproduces with
      # or without this drain (profiles/two_process_stall.md); gloo on device tensors is a
      # test / rehearsal configuration, so the conservative order is kept: compute first, then
      # the collective (DTF_GLOO_DRAIN=0 overlaps them).
      torch.cuda.current_stream(g.device).synchronize()
    cs = self._own_stream(g.device) if g.is_cuda else None
    if cs is not None:
      # stream-ordered, nothing blocks the host: the communication stream waits for the
      # gradients enqueued so far, the collective is issued as a blocking-API call *on that
      # stream*, and finish() makes the compute stream wait for the communication stream
      cs.wait_stream(torch.cuda.current_stream(g.device))
      ex = self._executor
      if ex is not None:
        if self._side is None:
          self._side = ex.side_stream()          # None until the side lane has been used
        if self._side is not None:
          cs.wait_stream(self._side)             # weight gradients of this bucket
      with torch.cuda.stream(cs):
        self._issue(g, async_op=False)
      self._stream_pending = True
    else:
      self._pending.append(self._issue(g, async_op=True))

  def _mesh(self, g: torch.Tensor):
    """The all-link exchange over `torch.distributed` point-to-point calls (the native
    communicator runs the same schedule from C++, `RcclComm::all_reduce_mesh`): blocking."""
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    n = self.n
    ranks = list(range(n)) if self.group is None else dist.get_process_group_ranks(self.group)
    me = ranks.index(self.comm.rank)
    shard = g.numel() // n
    scratch = torch.empty((n - 1, shard), dtype=g.dtype, device=g.device)
    ops = []
    for j in range(n):
      if j == me:
        continue
      slot = j if j < me else j - 1
      ops.append(dist.P2POp(dist.isend, g[j * shard:(j + 1) * shard], ranks[j], self.group))
      ops.append(dist.P2POp(dist.irecv, scratch[slot], ranks[j], self.group))
    for w in dist.batch_isend_irecv(ops):
      w.wait()
    own = g[me * shard:(me + 1) * shard]
    own.add_(scratch.sum(0))
    ops = []
    for j in range(n):
      if j == me:
        continue
      ops.append(dist.P2POp(dist.isend, own, ranks[j], self.group))
      ops.append(dist.P2POp(dist.irecv, g[j * shard:(j + 1) * shard], ranks[j], self.group))
    for w in dist.batch_isend_irecv(ops):
      w.wait()
    return None

  def finish(self):
    """Orders the optimizer after every outstanding collective; once per step this is also where
    an asynchronous RCCL error surfaces (the watchdog thread catches it between steps)."""
    for w in self._pending:
      if w is not None:
        w.wait()
    self._pending = []
    if self._stream_pending:
      torch.cuda.current_stream(self._stream.device).wait_stream(self._stream)
      self._stream_pending = False
    if self._native:
      self._native.check()

  def reduce_all_now(self):
    """Non-overlapped path: reduce every bucket after backward has finished."""
    for b in self.buckets:
      self.on_bucket_ready(b)
    self.finish()
