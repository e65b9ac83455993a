"""The engine's own RCCL communicator (`ops/csrc/rccl_comm.cpp`) behind a small Python handle.

This is the data plane of GPU jobs (`--comm native`, the default when the job runs on RCCL;
`--comm torch` / `DTF_COMM=torch` selects `torch.distributed`, `parallel/comm.py`, which also
stays the control plane: rendezvous store, barriers, scalar reductions).  Calls are issued from
the extension on a HIP stream handle, can be recorded as launch-plan entries (op kind
`allreduce`) and are therefore captured into the step's hipGraph.  Algorithms: `oneshot`
(`ncclAllReduce`, RCCL picks ring / tree / direct itself: the default), `direct` (reduce-scatter
+ all-gather, `--all_reduce_alg ring`) and `mesh` (`--all_reduce_alg mesh`): an all-link
exchange written for xGMI's full mesh of point-to-point links (7 x ~153 GB/s per GPU) - grouped
sends / receives move 1/N shards over every link at once, where a ring drives one link per
direction.  The parameter server's push / pull transfers use `send` / `recv` on two further
communicators, one per direction (`parallel/parameter_server.py`).

Failures: a watchdog thread polls `ncclCommGetAsyncError` once a second; on an error it aborts
the communicator (`ncclCommAbort`, so kernels blocked in a collective return) and ends the
process with exit code 70, which the launcher turns into the end of the whole job
(`launch.py`).  `check()` is the same poll on the caller's thread, once per step.

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

  def __init__(self, rank: int, world_size: int, store=None, device: Optional[torch.device] = None,
               watchdog: bool = True):
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
    # `auto` = RCCL's own all-reduce unless a threshold asks for reduce-scatter + all-gather
    min_mb = os.environ.get("DTF_ALLREDUCE_DIRECT_MIN_MB")
    self._c.direct_min_bytes = int(float(min_mb) * (1 << 20)) if min_mb is not None else 1 << 62
    if watchdog and os.environ.get("DTF_COMM_WATCHDOG", "1") != "0":
      self._c.start_watchdog(int(os.environ.get("DTF_COMM_WATCHDOG_MS", "1000")), self.EXIT_CODE)
    logging.info("native RCCL communicator: rank %d / %d%s", self.rank, self.world_size,
                 ", watchdog on" if watchdog else "")

  EXIT_CODE = 70        # EX_SOFTWARE: what a rank exits with when its communicator failed

  def check(self):
    """Raises when RCCL has reported an asynchronous error (the communicator is aborted by then)."""
    err = self._c.async_error()
    if err:
      raise RuntimeError(err)

  def close(self):
    self._c.stop_watchdog()

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
    elif mode == "mesh" and n % self.world_size == 0 and dt == 2:
      self._c.all_reduce_mesh(p, n, dt, s)
    else:
      self._c.all_reduce_auto(p, n, dt, s)

  # -- point to point (parameter server) ----------------------------------------------------
  def send(self, t: torch.Tensor, peer: int, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.send(p, n, dt, int(peer), s)

  def recv(self, t: torch.Tensor, peer: int, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.recv(p, n, dt, int(peer), s)

  def group(self):
    """Context manager: the sends / receives issued inside form one RCCL group (they progress
    together; a send and the matching receive of one rank to itself must share a group)."""
    return _Group(self._c)

  def broadcast(self, t: torch.Tensor, root: int = 0, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.broadcast(p, n, dt, root, s)

  def reduce(self, t: torch.Tensor, root: int = 0, stream=None):
    p, n, dt, s = self._args(t, stream)
    self._c.reduce(p, n, dt, root, s)

  def plan_entry(self, t: torch.Tensor, mode: str = "auto") -> dict:
    """Descriptor of an `allreduce` launch-plan entry for `t` (captured with the plan)."""
    p, n, dt, _ = self._args(t, 0)
    if mode == "mesh" and (n % self.world_size or dt != 2):
      mode = "auto"
    return dict(comm=self.handle, buf=p, count=n, dt=dt,
                mode={"oneshot": 0, "direct": 1, "auto": 2, "mesh": 3}[mode])


class _Group:
  def __init__(self, c):
    self._c = c

  def __enter__(self):
    self._c.group_start()
    return self

  def __exit__(self, *exc):
    self._c.group_end()
    return False


def _probe(nc: NativeComm, seconds: float) -> bool:
  """One small all-reduce on a stream of its own, polled from the host: True when it completed
  within `seconds` with the right sum."""
  import time  # pylint: disable=import-outside-toplevel
  with torch.cuda.device(nc.device):
    t = torch.ones(1024, device=nc.device, dtype=torch.float32)
    s = torch.cuda.Stream(nc.device)
    s.wait_stream(torch.cuda.current_stream(nc.device))
    nc.all_reduce(t, "oneshot", stream=s.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(s)
    deadline = time.time() + seconds
    while not ev.query():
      if time.time() > deadline:
        return False
      time.sleep(0.005)
    return bool((t == float(nc.world_size)).all().item())


def _agree(store, key: str, rank: int, world: int, ok: bool) -> bool:
  """True when every rank reports `ok` - through the rendezvous store, which needs no GPU."""
  store.set("%s/%d" % (key, rank), b"1" if ok else b"0")
  return all(bytes(store.get("%s/%d" % (key, r))) == b"1" for r in range(world))


def get_native_comm(comm) -> Optional[NativeComm]:
  """The process-wide native communicator over the ranks of `comm` (a `parallel.comm.Communicator`),
  created on first use; None when the job runs on the CPU.

  A communicator of several ranks is proven before it carries gradients: a small all-reduce must
  complete with the right sum within `DTF_COMM_PROBE_S` seconds (default 60) on EVERY rank - the
  verdicts are compared through the rendezvous store.  If any rank fails, all of them abort
  their communicator and get None: the job continues on `torch.distributed` instead of hanging
  in its first step (`DTF_COMM=native` turns the fallback into an error)."""
  global _NATIVE
  if _NATIVE is None:
    if comm.device.type != "cuda":
      return None
    store = comm.store() if (comm.world_size > 1 or comm.rehearse) else None
    nc = NativeComm(comm.rank, comm.world_size, store if comm.world_size > 1 else None,
                    comm.device)
    if comm.world_size > 1:
      seconds = float(os.environ.get("DTF_COMM_PROBE_S", "60"))
      ok = _probe(nc, seconds) if seconds > 0 else True
      if not _agree(store, "dtf/rccl_probe/%d" % _SEQ, comm.rank, comm.world_size, ok):
        logging.warning("native RCCL communicator failed its start-up all-reduce on %s: aborted",
                        "this rank" if not ok else "another rank")
        nc.close()
        nc._c.abort()  # pylint: disable=protected-access
        _NATIVE = False
        raise RuntimeError("the native RCCL communicator did not pass its start-up probe")
    _NATIVE = nc
  if _NATIVE is False:
    raise RuntimeError("the native RCCL communicator did not pass its start-up probe")
  return _NATIVE


def new_native_comm(comm, watchdog: bool = True) -> NativeComm:
  """A further communicator over the ranks of `comm` (COLLECTIVE: every rank calls it, in the
  same order): the parameter server keeps one per transfer direction, each with its own stream,
  so that a receive waiting for its sender never holds up traffic the other way."""
  store = comm.store() if comm.world_size > 1 else None
  return NativeComm(comm.rank, comm.world_size, store, comm.device, watchdog=watchdog)


def comm_choice(backend: str, device_type: str, requested: Optional[str] = None) -> str:
  """`native` or `torch` for the gradient collectives of a job: what `--comm` / `DTF_COMM` ask
  for, else the engine's communicator whenever the job runs on RCCL with the extension built."""
  want = (requested or "auto").lower()
  if want == "auto":                   # the flag left at its default: the environment may decide
    want = (os.environ.get("DTF_COMM") or "auto").lower()
  if want not in ("auto", "native", "torch"):
    raise ValueError("--comm must be native, torch or auto (got %r)" % want)
  if backend != "nccl" or device_type != "cuda":
    return "torch"
  if want == "torch":
    return "torch"
  from ..ops import native  # pylint: disable=import-outside-toplevel
  if not native.available():
    if want == "native":
      raise RuntimeError("--comm native needs the in-tree extension (python -m "
                         "distributed_tensorflow_amd.ops.build)")
    return "torch"
  return "native"
