"""One log record with everything a first multi-GPU run needs to be diagnosed from its log alone:
the effective HIP / RCCL environment, the process group, the gradient reducer's data plane,
algorithm, buckets and stream, and how the executor launches its step."""
from __future__ import annotations

import os
from typing import List, Optional

__all__ = ["runtime_report", "log_runtime_report"]

# environment that changes how streams, queues and inter-process transfers behave
_ENV = ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES",
        "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "NCCL_DEBUG", "NCCL_ALGO", "NCCL_PROTO",
        "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE", "HSA_NO_SCRATCH_RECLAIM",
        "TORCH_NCCL_ASYNC_ERROR_HANDLING", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE",
        "LOCAL_WORLD_SIZE")


def runtime_report(strategy=None, model=None) -> List[str]:
  """Lines describing the effective configuration of this rank."""
  import torch  # pylint: disable=import-outside-toplevel
  lines = []
  env = ["%s=%s" % (k, os.environ[k]) for k in _ENV if k in os.environ]
  env += ["%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DTF_")]
  lines.append("environment: " + (" ".join(env) if env else "(defaults)"))
  lines.append("torch %s, hip %s, devices visible: %d" % (
      torch.__version__, getattr(torch.version, "hip", None),
      torch.cuda.device_count() if torch.cuda.is_available() else 0))
  comm = getattr(strategy, "comm", None)
  if comm is not None:
    lines.append("process group: backend %s, rank %d of %d, device %s%s" % (
        comm.backend, comm.rank, comm.world_size, comm.device,
        ", one-rank rehearsal" if getattr(comm, "rehearse", False) else ""))
  elif strategy is None:
    lines.append("process group: none (single process)")
  if strategy is not None:
    lines.append("strategy: %s, synchronous %s, replicas %d" % (
        strategy.name, strategy.synchronous, strategy.num_replicas_in_sync))
    describe = getattr(strategy, "describe", None)
    if describe is not None:
      lines.append("parameter server: " + describe())
  if model is not None:
    red = None
    try:
      red = model._grad_reducer()  # pylint: disable=protected-access
    except Exception as e:  # pylint: disable=broad-except
      lines.append("gradient reducer: unavailable (%s)" % e)
    if red is not None:
      lines.append("gradient reducer: " + red.describe())
    elif strategy is not None and strategy.synchronous:
      lines.append("gradient reducer: none (one replica)")
    for n, ex in getattr(model, "_executors", {}).items():
      mode = {True: "hipGraph replay", False: "eager launch from the native plan",
              None: "undecided (timed in the first 8 steps)"}.get(getattr(ex, "use_graph", None), "n/a")
      lines.append("executor: %s, batch %d, dtype %s, step launch: %s" % (
          type(ex).__name__, n, getattr(model, "dtype", "?"), mode))
  return lines


def log_runtime_report(strategy=None, model=None, rank: Optional[int] = None, stream=None):
  """Rank 0 writes the report (to `stream`, default stderr, so that it survives any verbosity)."""
  import sys  # pylint: disable=import-outside-toplevel
  if rank is None:
    rank = getattr(strategy, "rank", 0) if strategy is not None else 0
  if rank != 0:
    return
  out = stream or sys.stderr
  for line in runtime_report(strategy, model):
    out.write("[dtf runtime] %s\n" % line)
  out.flush()
