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
only ever posts a transfer that the requesting worker posts too, and a worker
posts its transfers in the order of its requests.

Requests:

* `pull:<rank>:<fmt>[:fresh]` - PS -> worker: the parameter shard as fp32 or bfloat16
  (a 16-bit worker would round the fp32 values the same way itself; half the
  bytes), PS 0 also the BatchNorm moving statistics.
* `push:<rank>:<lr>:<divisor>:<fmt>:<lo>-<hi>,...` - worker -> PS, protocol `stream`: the
  whole push announced before the backward pass; the gradient shard follows as pieces (one per
  gradient bucket that intersects the shard, in the order the backward pass completes them),
  fp32 or bfloat16; to PS 0 also the moving statistics; last a two-element trailer
  `[skip, factor]`: skip != 0 - the worker found a non-finite gradient (dynamic loss scaling),
  drop the push; factor multiplies the gradients (1 / loss scale when the scale lives on the
  worker's GPU).
* `piece:<rank>:<fmt>:<lo>-<hi>` and `commit:<rank>:<lr>:<divisor>:<fmt>` - protocol `pieces`:
  a bucket is announced when it is *ready to leave* (its kernels have finished), so a PS only
  ever posts a receive whose data is already waiting at the sender; `commit` closes the push:
  moving statistics (PS 0) and the trailer follow, and the PS applies it.
* `done:<rank>` - the worker is finished; a PS returns once every worker said so
  (a clean shutdown instead of `pkill`).

Overlap.  A worker's step is

    wait for the prefetched parameters -> request the *next* step's parameters
    -> forward / backward, every gradient bucket leaving as soon as it is complete
    -> trailer

with the transfers on two lane threads (one receiving, one sending, each a FIFO)
so that the host thread only enqueues: the next parameters arrive while this
step computes (they miss this step's own update: staleness + 1, on top of the
staleness asynchronous SGD has anyway; `DTF_PS_PREFETCH=0` restores strict
pull -> compute -> push), and all but the last bucket of the push is hidden
behind the backward pass.

Data planes (`ps_transport.py`).  Over gloo (`stream` protocol) the PS posts the receives of a
whole push and moves on: up to two pushes are in flight into two buffers while it serves other
requests.  Over RCCL a posted receive is a kernel that occupies its stream until the sender
shows up, so the RCCL plane keeps one communicator and stream per direction and runs the
`pieces` protocol: nothing is posted before its data exists, every worker lands in a buffer of
its own, and the PS side is stream-ordered end to end - receive, fused SGD-momentum+L2 over the
shard with the trailer read *on the device* (no host synchronisation per push), the cast of the
next pull behind the update - so its request loop never waits for the GPU.
"""
from __future__ import annotations

import collections
import concurrent.futures
import datetime
import os
import time
from typing import Deque, Dict, List, Optional, Tuple

import torch

from ..utils import logging
from .cluster import TaskInfo, resolve_task
from .comm import Communicator, get_communicator
from .ps_transport import PULL, PUSH, make_plane
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
               poll_timeout_s: float = 1800.0, comm_kind: Optional[str] = None):
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
    self.updates_dropped = 0
    self._host: Dict = {}
    self.prefetch = os.environ.get("DTF_PS_PREFETCH", "1") != "0"
    # gradient buckets of this size leave while the backward pass is still running; what is
    # left for after the step is the small tail bucket of the first layers
    self.bucket_bytes = int(float(os.environ.get("DTF_PS_BUCKET_MB", "16")) * (1 << 20))
    # gradients travel as bfloat16 when the worker computes in a 16-bit type, unless told otherwise
    self.push_dtype = os.environ.get("DTF_PS_PUSH_DTYPE", "auto").lower()
    # The data plane (COLLECTIVE: every rank builds it here).  gloo moves one message with one
    # progress thread per process group (2.5 GB/s on the test host, and no faster in both
    # directions at once), so its transfers are striped over several groups, half of them for
    # each direction (51 MB each way: 40.7 -> 20.5 ms with 4); the RCCL plane has a communicator
    # and a stream per direction.
    probe = None
    if comm.device.type == "cuda" and task.job != "ps":
      probe = torch.cuda.default_stream(comm.device)
    self.plane = make_plane(comm, comm_kind, probe_against=probe)
    # `pieces`: buckets are announced when they are ready to leave (the only protocol in which a
    # PS never posts a receive that has to wait: required by RCCL, optional elsewhere);
    # `stream`: the whole push is announced up front and the PS keeps two pushes in flight
    # (gloo); `whole`: one announcement after the step (torch.distributed on RCCL).
    default = "pieces" if self.plane.device_side else \
        ("stream" if comm.backend != "nccl" else "whole")
    self.protocol = os.environ.get("DTF_PS_PROTOCOL", default).lower()
    if self.protocol not in ("pieces", "stream", "whole"):
      raise ValueError("DTF_PS_PROTOCOL must be pieces, stream or whole")
    if self.plane.device_side and self.protocol != "pieces":
      raise ValueError("the RCCL data plane runs the `pieces` protocol")
    if os.environ.get("DTF_PS_STREAM_PUSH", "1") == "0" and self.protocol == "stream":
      self.protocol = "whole"
    self.stream_push = self.protocol == "stream"
    self._recv_lane = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix="ps-recv")
    self._send_lane = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix="ps-send")
    self._prefetched = None        # (future, fmt) of the parameters requested ahead
    self._push: Optional[Dict] = None
    self._last_send = None         # future of the previous step's last transfer
    self._copy_stream = None
    self._executor = None          # set by orders_side_lane()
    self._h2d_done = None
    # seconds the training thread spent waiting for transfers: what is *not* hidden
    self.wait_s = {"pull": 0.0, "push": 0.0, "steps": 0, "gpu_ms": 0.0, "host_s": 0.0}
    self._step_events = None
    self._t_begin = None

  def describe(self) -> str:
    return "%d ps + %d workers, data plane: %s, protocol %s, push buckets %.0f MB, prefetch %s" % (
        self.num_ps, self.num_workers, self.plane.describe(), self.protocol,
        self.bucket_bytes / 2 ** 20, "on" if self.prefetch else "off")

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

  def _request_waiting(self, ps: int, head: int) -> bool:
    try:
      return bool(self._store.check(["ps%d/req/%d" % (ps, head)]))
    except Exception:  # pragma: no cover - a store without check(): behave as if busy
      return True

  # -- transfers (see ps_transport.py) ------------------------------------------------------------
  def _isend(self, t: torch.Tensor, dst: int, direction: int) -> List:
    return self.plane.isend(t, dst, direction)

  def _irecv(self, t: torch.Tensor, src: int, direction: int) -> List:
    return self.plane.irecv(t, src, direction)

  PULL, PUSH = PULL, PUSH        # directions: PS -> worker, worker -> PS

  # -- buffers -----------------------------------------------------------------------
  # RCCL moves device memory directly.  Any other backend (gloo: parameter server in host
  # memory like the reference's `/job:ps/.../CPU:0`, or a rehearsal on fewer GPUs than ranks)
  # gets device tensors staged through pinned host buffers here: gloo's own handling of device
  # tensors took seconds for the 102 MB of ResNet-50 on the test machine.
  def _staged(self, device) -> bool:
    return self.plane.staged(device)

  def _buf(self, tag, n: int, dtype, device=None, pinned=False) -> torch.Tensor:
    key = (tag, n, dtype, str(device), pinned)
    if key not in self._host:
      if pinned:
        self._host[key] = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
      else:
        self._host[key] = torch.empty(max(n, 1), dtype=dtype, device=device)
    return self._host[key]

  def _stream(self, device):
    """The stream of the staging copies.  HIP multiplexes streams onto a few hardware queues
    and two streams on one queue run strictly in submission order: a copy stream that shared
    the queue of the compute stream or of the executor's weight-gradient lane held their
    kernels up behind its waits and transfers (ResNet-50 step 21.6 -> 29 ms on the GPU).  The
    native extension hands out a stream it has probed to run concurrently with both."""
    if self._copy_stream is None:
      from ..ops import native  # pylint: disable=import-outside-toplevel
      if native.available():
        main = torch.cuda.default_stream(device)
        h = native.ext().concurrent_stream(int(main.cuda_stream), 0)
        self._copy_stream = torch.cuda.ExternalStream(h, device=device)
      else:
        self._copy_stream = torch.cuda.Stream(device)
    return self._copy_stream

  def orders_side_lane(self, executor) -> bool:
    """True when the pushes wait for the executor's side lane (weight gradients) themselves, on
    the copy stream: the executor may then skip the main-lane joins at bucket boundaries
    (`HipExecutor.train_step(soft_joins=True)`)."""
    if not hasattr(executor, "side_stream") or executor.device.type != "cuda":
      return False
    self._executor = executor
    return True

  # -- worker side -------------------------------------------------------------------
  def _shards(self, store):
    return shard_ranges(store.num_elements, self.num_ps)

  def flush(self):
    """Returns when every transfer this worker has announced has left (its last push included)."""
    if self._last_send is not None:
      self._last_send.result()

  def _start_pull(self, store, fmt: str, fresh: bool = False):
    """Posts the pull requests and queues the receives on the receiving lane; returns the
    future of the lane task, whose result is the list of filled landing buffers."""
    dtype = torch.bfloat16 if fmt == "bf16" else torch.float32
    shards = self._shards(store)
    dev = store.device
    staged = self._staged(dev)
    for k in range(self.num_ps):
      self._post(k, "pull:%d:%s%s" % (self.comm.rank, fmt, ":fresh" if fresh else ""))
    consumed = self._h2d_done

    def task():
      if dev.type == "cuda":
        torch.cuda.set_device(dev)
      if consumed is not None:
        # the previous pull has been copied out of the landing buffers before they are refilled
        if self.plane.device_side:
          self.plane.order_after(self.PULL, consumed)
        else:
          consumed.synchronize()
      out = []
      for k, (a, b) in enumerate(shards):
        land, works = None, []
        if b > a:
          land = self._buf(("pull", k), b - a, dtype, None if staged else dev, pinned=staged)
          works = self._irecv(land[:b - a], k, self.PULL)
        state = None
        if k == 0 and store.num_state_elements:
          state = self._buf("pull_state", store.num_state_elements, torch.float32,
                            None if staged else dev, pinned=staged)
          works += self._irecv(state[:store.num_state_elements], 0, self.PULL)
        for w in works:
          w.wait()
        if staged and land is not None:
          # host -> device on the copy stream, behind the compute of the running step: the
          # training thread finds the parameters in device memory
          stream = self._stream(dev)
          with torch.cuda.stream(stream):
            tmp = self._buf(("pull_dev", k), b - a, land.dtype, dev)
            tmp[:b - a].copy_(land[:b - a], non_blocking=True)
            if state is not None:
              stmp = self._buf("pull_state_dev", store.num_state_elements, torch.float32, dev)
              stmp.copy_(state[:store.num_state_elements], non_blocking=True)
              state = stmp
          stream.synchronize()
          land = tmp
        out.append((a, b, land, state))
      return out

    return self._recv_lane.submit(task)

  def pull(self, store, half: bool = False, fresh: bool = False):
    """Leaves the PS ranks' parameters (and moving statistics) in `store`.

    `half`: the worker computes in a 16-bit format, so it asks for bfloat16 parameters - half
    the bytes per pull, and exactly the values its executor would round the fp32 ones to.  The
    master copy stays fp32 on the PS (updates are applied there).  Parameters requested ahead by
    the previous step are used when there are any; `fresh` discards them, waits until this
    worker's own pushes have left and asks again (evaluation, checkpoints)."""
    fmt = "bf16" if half else "fp32"
    pre, self._prefetched = self._prefetched, None
    if pre is not None and (fresh or pre[1] != fmt):
      pre[0].result()            # the PS is sending it anyway: receive and drop
      pre = None
    if pre is None:
      self.flush()
      pre = (self._start_pull(store, fmt, fresh), fmt)
    t0 = time.time()
    landed = pre[0].result()
    self.wait_s["pull"] += time.time() - t0
    dev = store.device
    for a, b, land, state in landed:
      if land is not None:
        store.params[a:b].copy_(land[:b - a])       # device to device, widening bf16
      if state is not None:
        store.state.copy_(state[:store.num_state_elements], non_blocking=True)
    if dev.type == "cuda":
      self._h2d_done = torch.cuda.Event()
      self._h2d_done.record()
    store.version += 1

  def begin_push(self, store, buckets, lr: float, divisor: float = 1.0, half: bool = False):
    """Announces this step's gradient push (and the next step's pull) before the backward pass;
    `on_bucket_ready` then hands each finished bucket to the sending lane."""
    if self._last_send is not None:
      t0 = time.time()
      self._last_send.result()      # the previous push has left the gradient / staging buffers
      self.wait_s["push"] += time.time() - t0
      self._last_send = None
    self.wait_s["steps"] += 1
    if self._step_events is not None:       # complete by now: the sends waited for the GPU
      self.wait_s["gpu_ms"] += self._step_events[0].elapsed_time(self._step_events[1])
      self._step_events = None
    if store.device.type == "cuda":
      self._stream(store.device)            # created (and probed) on the training thread
      self._step_events = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
      self._step_events[0].record()
    self._t_begin = time.time()
    if self.prefetch:
      self._prefetched = (self._start_pull(store, "bf16" if half else "fp32"),
                          "bf16" if half else "fp32")
    fmt = self.push_dtype
    if fmt not in ("bf16", "fp32"):
      fmt = "bf16" if half else "fp32"
    order = sorted(buckets, key=lambda bk: bk.index)
    pieces: List[List[Tuple[int, int]]] = []       # per PS: element ranges in sending order
    for a, b in self._shards(store):
      pieces.append([(max(a, bk.start), min(b, bk.end)) for bk in order
                     if min(b, bk.end) > max(a, bk.start)])
    self._push = dict(store=store, fmt=fmt, order=order, pieces=pieces, next=0, ready={},
                      lr=float(lr), divisor=float(divisor), posted=False, works=[], keep=[])
    if self.stream_push:
      self._post_push()

  def _post_push(self):
    p = self._push
    shards = self._shards(p["store"])
    for k, (a, _b) in enumerate(shards):
      self._post(k, "push:%d:%r:%r:%s:%s" % (
          self.comm.rank, p["lr"], p["divisor"], p["fmt"],
          ",".join("%d-%d" % (lo - a, hi - a) for lo, hi in p["pieces"][k])))
    p["posted"] = True

  def on_bucket_ready(self, bucket):
    """Called by the executor when the kernels that complete `bucket` have been enqueued."""
    p = self._push
    ev = None
    if p["store"].device.type == "cuda":
      ev = [torch.cuda.Event()]
      ev[0].record()
      side = self._executor.side_stream() if self._executor is not None else None
      if side is not None:               # the bucket's weight gradients
        ev.append(torch.cuda.Event())
        ev[1].record(side)
    p["ready"][bucket.index] = ev
    # buckets leave strictly in the announced order
    while p["next"] < len(p["order"]) and p["order"][p["next"]].index in p["ready"]:
      bk = p["order"][p["next"]]
      p["next"] += 1
      if self.protocol in ("stream", "pieces"):
        self._queue_bucket(p, bk, p["ready"][bk.index])

  def _queue_bucket(self, p, bk, ev):
    store, fmt = p["store"], p["fmt"]
    dev = store.device
    staged = self._staged(dev)
    dtype = torch.bfloat16 if fmt == "bf16" else torch.float32
    shards = self._shards(store)
    n = store.num_elements
    pieces = self.protocol == "pieces"
    device_side = self.plane.device_side

    def announce(k, a, lo, hi):
      if pieces:              # the data is ready: the PS may post its receive without waiting
        self._post(k, "piece:%d:%s:%d-%d" % (self.comm.rank, fmt, lo - a, hi - a))

    def task():
      src = store.grads
      if dev.type == "cuda":
        torch.cuda.set_device(dev)
        stream = self._stream(dev)
        with torch.cuda.stream(stream):
          for e in ev or []:
            stream.wait_event(e)
          if fmt == "bf16":
            half = self._buf("push_half", n, dtype, dev)
            half[bk.start:bk.end].copy_(store.grads[bk.start:bk.end])
            src = half
          if staged:
            host = self._buf("push_host", n, dtype, None, pinned=True)
            host[bk.start:bk.end].copy_(src[bk.start:bk.end], non_blocking=True)
            src = host
          if staged or pieces:
            stream.synchronize()          # the bucket (and its 16-bit copy) exist: announce it
          if device_side:
            self.plane.order_after(self.PUSH, stream)
          for k, (a, b) in enumerate(shards):
            lo, hi = max(a, bk.start), min(b, bk.end)
            if hi > lo:
              announce(k, a, lo, hi)
              p["works"] += self._isend(src[lo:hi], k, self.PUSH)
        return
      if fmt == "bf16":
        src = store.grads[bk.start:bk.end].to(dtype)
        p["keep"].append(src)
        for k, (a, b) in enumerate(shards):
          lo, hi = max(a, bk.start), min(b, bk.end)
          if hi > lo:
            announce(k, a, lo, hi)
            p["works"] += self._isend(src[lo - bk.start:hi - bk.start], k, self.PUSH)
        return
      for k, (a, b) in enumerate(shards):
        lo, hi = max(a, bk.start), min(b, bk.end)
        if hi > lo:
          announce(k, a, lo, hi)
          p["works"] += self._isend(src[lo:hi], k, self.PUSH)

    self._last_send = self._send_lane.submit(task)

  def end_push(self, finite: bool = True, trailer: Optional[torch.Tensor] = None):
    """After the step: whatever has not left yet, the moving statistics (to PS 0) and the
    trailer `[skip, factor]` that tells every PS whether to apply this push and what to multiply
    the gradients by.  `trailer`: two floats on the worker's device, written by the device-side
    loss-scale kernel (the host never learns whether the step was finite); else `finite` says it."""
    p = self._push
    store = p["store"]
    dev = store.device
    ev = None
    self.wait_s["host_s"] += time.time() - self._t_begin
    if dev.type == "cuda":
      side = self._executor.side_stream() if self._executor is not None else None
      if side is not None:
        # the executor skipped its joins: the step ends when the weight-gradient lane has
        torch.cuda.current_stream(dev).wait_stream(side)
      if self.plane.device_side and store.num_state_elements:
        # the moving statistics as they are now: the next step's forward pass updates them
        # while this push may still be waiting for its turn
        snap = self._buf("push_state_dev", store.num_state_elements, torch.float32, dev)
        snap.copy_(store.state)
      ev = [torch.cuda.Event()]
      ev[0].record()
      self._step_events[1].record()
    if self.protocol == "whole":
      self._post_push()                       # announced only now, sent in one go
      for bk in p["order"]:
        self._queue_bucket(p, bk, ev)
    else:
      while p["next"] < len(p["order"]):      # an executor that never called back
        self._queue_bucket(p, p["order"][p["next"]], ev)
        p["next"] += 1
    staged = self._staged(dev)
    device_side = self.plane.device_side
    pieces = self.protocol == "pieces"
    if trailer is None:
      trailer = torch.tensor([0.0 if finite else 1.0, 1.0], dtype=torch.float32,
                             device="cpu" if (staged or dev.type != "cuda") else dev)

    def task():
      state = store.state
      if device_side and store.num_state_elements:
        state = self._buf("push_state_dev", store.num_state_elements, torch.float32, dev)
      tr = trailer
      if dev.type == "cuda":
        torch.cuda.set_device(dev)
        for e in ev or []:
          e.synchronize()
        if staged:
          tr = trailer.cpu()
          if store.num_state_elements:
            host = self._buf("push_state", store.num_state_elements, torch.float32, None, pinned=True)
            host.copy_(store.state)
            state = host
      works = p["works"]
      for k in range(self.num_ps):
        if pieces:
          self._post(k, "commit:%d:%r:%r:%s" % (self.comm.rank, p["lr"], p["divisor"], p["fmt"]))
        if k == 0 and store.num_state_elements:
          works += self._isend(state[:store.num_state_elements], 0, self.PUSH)
        works += self._isend(tr, k, self.PUSH)
      # The sends to different PS ranks are independent of each other (none of them is waited
      # for before all are posted): a PS that is busy completing another worker's push cannot
      # hold up what this worker owes to the other PS ranks, which rules out a cycle of waits
      # across several PS ranks and workers.
      for w in works:
        w.wait()
      if device_side:
        self.plane.check()
      p["keep"].clear()

    self._last_send = self._send_lane.submit(task)
    self._push = None

  def push(self, store, lr: float, divisor: float = 1.0, buckets=None, half: bool = False,
           finite: bool = True):
    """The whole gradient in one call (no overlap with a backward pass); returns when it has
    left."""
    from ..engine.params import Bucket  # pylint: disable=import-outside-toplevel
    keep, self.prefetch = self.prefetch, False
    try:
      self.begin_push(store, buckets or [Bucket(0, 0, store.num_elements, [])], lr, divisor, half)
    finally:
      self.prefetch = keep
    self.end_push(finite)
    self._last_send.result()

  def side_lane(self):
    """The stream of the executor's weight-gradient lane when the pushes order themselves behind
    it (`orders_side_lane`), else None."""
    return self._executor.side_stream() if self._executor is not None else None

  def shutdown(self):
    if self.is_ps_task or self._done_sent or self._store is None:
      return
    if self._prefetched is not None:          # the PS is sending it: take delivery
      self._prefetched[0].result()
      self._prefetched = None
    if self._last_send is not None:
      self._last_send.result()
    for k in range(self.num_ps):
      self._post(k, "done:%d" % self.comm.rank)
    self._done_sent = True
    n = max(self.wait_s["steps"], 1)
    logging.info("parameter-server worker %d: %d steps, per step %.2f ms waiting for parameters, "
                 "%.2f ms for the previous push to leave, %.2f ms enqueueing the step, %.2f ms of "
                 "GPU time from its first to its last kernel", self.worker_index,
                 self.wait_s["steps"], 1e3 * self.wait_s["pull"] / n, 1e3 * self.wait_s["push"] / n,
                 1e3 * self.wait_s["host_s"] / n, self.wait_s["gpu_ms"] / n)

  # -- PS side ---------------------------------------------------------------------
  def serve(self, model) -> int:
    """Request loop of a PS rank. Returns the number of updates applied."""
    store = model.store
    opt = model.optimizer
    k = self.task.task_index
    a, b = self._shards(store)[k]
    n = b - a
    dev = store.device
    staged = self._staged(dev)
    device_side = self.plane.device_side
    shard = _ShardView(store, a, b)
    has_state = k == 0 and store.num_state_elements > 0
    ns = store.num_state_elements
    head = 0
    active = self.num_workers
    inflight: Deque[Dict] = collections.deque()
    free_slots = [1, 0]
    open_push: Dict[int, Dict] = {}          # `pieces` protocol: worker rank -> push being received
    logging.info("parameter server %d/%d serving shard [%d, %d) for %d workers (%s, protocol %s%s)",
                 k, self.num_ps, a, b, active, self.plane.describe(), self.protocol,
                 ", host-staged" if staged else "")
    t0 = time.time()

    def land(tag, slot, count, dtype):
      return self._buf((tag, slot), count, dtype, None if staged else dev, pinned=staged)

    busy = {"recv_wait": 0.0, "apply": 0.0, "pull_stage": 0.0, "pull_send": 0.0, "idle": 0.0}
    main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
    # device-side plane: [updates applied, updates dropped] counted on the GPU, events that order
    # the next pull's cast behind the last update and the next update behind the last cast
    counters = torch.zeros(2, dtype=torch.float32, device=dev) if device_side else None
    last = {"apply": None, "cast": None}

    def finalize(p):
      t1 = time.time()
      if device_side:
        # stream-ordered: the update waits for the receives on the device, the host moves on
        ev = torch.cuda.Event()
        ev.record(self.plane.stream(self.PUSH))
        main.wait_event(ev)
        if last["cast"] is not None:
          main.wait_event(last["cast"])        # a pull in flight still reads the old parameters
      else:
        for w in p["works"]:
          w.wait()
      t2 = time.time()
      busy["recv_wait"] += t2 - t1
      self._finalize(p, shard, opt, n, ns, has_state, staged, dev, counters)
      if device_side:
        last["apply"] = torch.cuda.Event()
        last["apply"].record(main)
        # the worker's next pieces land in the same buffer: not before this update has read it
        self.plane.order_after(self.PUSH, last["apply"])
      busy["apply"] += time.time() - t2
      if p.get("slot") is not None:
        free_slots.append(p["slot"])

    while active > 0 or inflight:
      if inflight and (active == 0 or not self._request_waiting(k, head)):
        finalize(inflight.popleft())          # nothing else to do: complete the oldest push
        continue
      t1 = time.time()
      msg = self._next_request(k, head)
      busy["idle"] += time.time() - t1
      head += 1
      parts = msg.split(":")
      op, src = parts[0], int(parts[1])
      if op == "pull":
        # read-your-writes for the requesting worker; other workers' pushes are applied first
        # only when they have arrived completely
        # (a `fresh` pull - evaluation, checkpoints - waits for every push that has been announced)
        fresh = len(parts) > 3 and parts[3] == "fresh"
        for p in list(inflight):
          if fresh or p["src"] == src or all(w.is_completed() for w in p["works"]):
            inflight.remove(p)
            finalize(p)
        dtype = torch.bfloat16 if (len(parts) > 2 and parts[2] == "bf16") else torch.float32
        t1 = time.time()
        sends = []
        if device_side:
          # cast and send on the pull stream, behind the last update; the host does not wait
          ps = self.plane.stream(self.PULL)
          if last["apply"] is not None:
            ps.wait_event(last["apply"])
          with torch.cuda.stream(ps):
            out = store.params[a:b]
            if n and dtype != torch.float32:
              half = self._buf("serve_half", n, dtype, dev)
              half[:n].copy_(out)
              out = half[:n]
            st_out = None
            if has_state:
              st_out = self._buf("serve_state_dev", ns, torch.float32, dev)
              st_out[:ns].copy_(store.state)
          last["cast"] = torch.cuda.Event()
          last["cast"].record(ps)
          if n:
            self._isend(out, src, self.PULL)
          if has_state:
            self._isend(st_out[:ns], src, self.PULL)
          busy["pull_stage"] += time.time() - t1
          continue
        if n:
          out = store.params[a:b]
          if dtype != torch.float32:
            half = self._buf("serve_half", n, dtype, dev)
            half[:n].copy_(out)
            out = half[:n]
          if staged:
            host = self._buf("serve_host", n, dtype, None, pinned=True)
            host[:n].copy_(out)                # synchronous device-to-host copy
            out = host[:n]
          sends = self._isend(out, src, self.PULL)
        t2 = time.time()
        busy["pull_stage"] += t2 - t1
        if has_state:
          st = store.state
          if staged:
            st = self._buf("serve_state", ns, torch.float32, None, pinned=True)
            st[:ns].copy_(store.state)
          sends += self._isend(st[:ns], src, self.PULL)
        for w in sends:
          w.wait()
        busy["pull_send"] += time.time() - t2
      elif op == "push":
        if not free_slots:
          finalize(inflight.popleft())
        slot = free_slots.pop()
        lr, divisor = float(parts[2]), float(parts[3])
        dtype = torch.bfloat16 if parts[4] == "bf16" else torch.float32
        grads = land("ps_land", slot, n, dtype)
        works = []
        for piece in (parts[5].split(",") if parts[5] else []):
          lo, hi = (int(v) for v in piece.split("-"))
          if not 0 <= lo < hi <= n:
            raise RuntimeError("parameter server: bad piece %s in %r" % (piece, msg))
          works += self._irecv(grads[lo:hi], src, self.PUSH)
        state = None
        if has_state:
          state = land("ps_state", slot, ns, torch.float32)
          works += self._irecv(state[:ns], src, self.PUSH)
        trailer = torch.zeros(2, dtype=torch.float32, device="cpu" if staged else dev)
        works += self._irecv(trailer, src, self.PUSH)
        p = dict(src=src, slot=slot, lr=lr, divisor=divisor, grads=grads, state=state,
                 trailer=trailer, works=works)
        if self.stream_push:
          inflight.append(p)
        else:
          finalize(p)
      elif op == "piece":
        # a bucket that is ready at the sender: the receive is posted at once (it never waits),
        # into the landing buffer that belongs to this worker
        dtype = torch.bfloat16 if parts[2] == "bf16" else torch.float32
        lo, hi = (int(v) for v in parts[3].split("-"))
        if not 0 <= lo < hi <= n:
          raise RuntimeError("parameter server: bad piece in %r" % msg)
        p = open_push.get(src)
        if p is None:
          p = open_push[src] = dict(src=src, slot=None, grads=land("ps_land", ("w", src), n, dtype),
                                    works=[], covered=0)
        if p["grads"].dtype != dtype:
          raise RuntimeError("parameter server: a push changed its format mid-way: %r" % msg)
        p["works"] += self._irecv(p["grads"][lo:hi], src, self.PUSH)
        p["covered"] += hi - lo
      elif op == "commit":
        p = open_push.pop(src, None)
        if p is None:                          # a shard without elements, or an empty push
          p = dict(src=src, slot=None, grads=land("ps_land", ("w", src), n, torch.float32),
                   works=[], covered=0)
        if p["covered"] != n:
          raise RuntimeError("parameter server: push of worker %d covers %d of %d elements"
                             % (src, p["covered"], n))
        p["lr"], p["divisor"] = float(parts[2]), float(parts[3])
        p["state"] = None
        if has_state:
          p["state"] = land("ps_state", ("w", src), ns, torch.float32)
          p["works"] += self._irecv(p["state"][:ns], src, self.PUSH)
        p["trailer"] = self._buf(("ps_trailer", src), 2, torch.float32,
                                 "cpu" if (staged or dev.type != "cuda") else dev)
        p["works"] += self._irecv(p["trailer"], src, self.PUSH)
        finalize(p)
      elif op == "done":
        active -= 1
      else:
        raise RuntimeError("parameter server: bad request %r" % msg)
    if open_push:
      raise RuntimeError("parameter server: workers %s finished with an open push" % sorted(open_push))
    if store.device.type == "cuda":
      torch.cuda.synchronize(store.device)
    if counters is not None:
      applied, dropped = (int(v) for v in counters.tolist())
      self.updates_applied += applied
      self.updates_dropped += dropped
      self.plane.check()
    logging.info("parameter server %d: %d updates applied, %d dropped in %.1f s; all workers done "
                 "(seconds: %s)", k, self.updates_applied, self.updates_dropped, time.time() - t0,
                 ", ".join("%s %.2f" % kv for kv in busy.items()))
    return self.updates_applied

  def _finalize(self, p, shard, opt, n, ns, has_state, staged, dev, counters=None):
    """Applies one completely received push (or drops it when its trailer `[skip, factor]` says
    so).  A trailer in device memory is never read by the host: the fused optimizer takes it as
    its skip flag / gradient factor, the moving statistics are selected on the device and the
    outcome is counted there."""
    store = shard.store
    tr = p["trailer"]
    momentum = getattr(opt, "momentum", 0.0)

    def widened(g):
      if g.device != dev:
        tmp = self._buf("ps_dev", n, g.dtype, dev)
        tmp[:n].copy_(g[:n], non_blocking=True)
        g = tmp
      if g.dtype != torch.float32:
        wide = self._buf("ps_f32", n, torch.float32, dev)
        wide[:n].copy_(g[:n])
        g = wide
      return g[:n]

    if tr.device.type == "cuda":
      if n:
        shard.apply_sgd(widened(p["grads"]), p["lr"], momentum, p["divisor"], skip=tr)
      applied = (tr[0] == 0)
      if has_state:
        store.state.copy_(torch.where(applied, p["state"][:ns], store.state))
      if counters is not None:
        counters[0] += applied
        counters[1] += ~applied
      return
    skip, factor = float(tr[0]), float(tr[1])
    if skip == 0:
      if n:
        shard.apply_sgd(widened(p["grads"]), p["lr"], momentum, p["divisor"] / factor)
      if has_state:
        store.state.copy_(p["state"][:ns], non_blocking=True)
      if staged:
        torch.cuda.current_stream(dev).synchronize()     # the landing buffers are free again
      self.updates_applied += 1
    else:
      self.updates_dropped += 1

  def barrier(self):
    pass   # asynchronous by design: no inter-worker barrier


class _ShardView:
  """The slice of the flat buffers one PS rank is authoritative for."""

  def __init__(self, store, a: int, b: int):
    self.store = store
    self.a, self.b = a, b

  def apply_sgd(self, grads: torch.Tensor, lr: float, momentum: float, divisor: float,
                skip: Optional[torch.Tensor] = None):
    """`skip`: two device floats [skip flag, gradient factor] read by the kernel itself."""
    from ..ops import fused  # pylint: disable=import-outside-toplevel
    st = self.store
    a, b = self.a, self.b
    fused.sgd_momentum_l2(st.params[a:b], grads, st.slot("momentum")[a:b],
                          st.l2_vector()[a:b], lr, momentum, 1.0 / divisor, skip)
    st.version += 1
