"""Cluster description: who am I, how many of us are there, where is rank 0.

The reference describes its cluster with a `TF_CONFIG` JSON literal written at
the top of every driver copy (`image_classification/resnet_cifar_main_dist.py:36-47`,
`ps_server/resnet_imagenet_main_dist_ps_0.py:40-50`) or with
`--worker_hosts/--task_index` resolved by `distribution_utils.configure_cluster`
(call site `resnet_imagenet_main.py:109-110`).  Here one driver per dataset is
enough: the role of a process comes from, in priority order,

1. the launcher environment `RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`
   (one process per GPU; what `torchrun` and `launch.py` export),
2. a `TF_CONFIG` environment variable with the reference's schema
   `{'cluster': {'ps': [...], 'chief': [...], 'worker': [...]}, 'task': {'type', 'index'}}`,
3. `--worker_hosts host:port,... --task_index i`.

A TF "worker" may own several GPUs (`--num_gpus`); each GPU becomes one rank, so
`world = n_tasks * gpus_per_task` and `rank = task_rank * gpus_per_task + local_rank`.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, List, Optional, Tuple

__all__ = ["ClusterSpec", "TaskInfo", "configure_cluster", "parse_tf_config",
           "resolve_task", "rendezvous_endpoint"]


@dataclasses.dataclass
class ClusterSpec:
  """Job name -> list of `host:port` strings (TF `ClusterSpec` analogue)."""
  jobs: Dict[str, List[str]]

  @property
  def num_workers(self) -> int:
    # The chief is a worker too (reference: configure_cluster counts chief + worker).
    return len(self.jobs.get("chief", [])) + len(self.jobs.get("worker", []))

  @property
  def num_ps(self) -> int:
    return len(self.jobs.get("ps", []))

  def task_address(self, job: str, index: int) -> str:
    return self.jobs[job][index]

  def ordered_tasks(self) -> List[Tuple[str, int]]:
    """Deterministic global order of tasks: ps first, then chief, then workers.

    Parameter-server tasks take the lowest ranks so that rank 0 is the shard
    holder in a 1 PS + N worker job, matching the reference's file numbering
    (`ps_server/resnet_imagenet_main_dist_ps_0.py` is the PS, `_k` is worker k-1).
    """
    out = []
    for job in ("ps", "chief", "worker"):
      out.extend((job, i) for i in range(len(self.jobs.get(job, []))))
    return out

  def as_dict(self):
    return {k: list(v) for k, v in self.jobs.items()}


@dataclasses.dataclass
class TaskInfo:
  """Resolved identity of this process."""
  job: str = "worker"            # 'worker' | 'chief' | 'ps'
  task_index: int = 0            # index inside the job
  rank: int = 0                  # global rank (one per GPU / process)
  local_rank: int = 0
  world_size: int = 1
  num_workers: int = 1           # data-parallel compute ranks
  num_ps: int = 0
  master_addr: str = "127.0.0.1"
  master_port: int = 29500
  cluster: Optional[ClusterSpec] = None

  @property
  def is_chief(self) -> bool:
    if self.job == "ps":
      return False
    if self.cluster and self.cluster.jobs.get("chief"):
      return self.job == "chief"
    return self.worker_rank == 0

  @property
  def worker_rank(self) -> int:
    """Rank among the compute workers (PS ranks come first)."""
    return self.rank - self.num_ps if self.job != "ps" else -1


def parse_tf_config(text: Optional[str] = None) -> Optional[dict]:
  text = os.environ.get("TF_CONFIG") if text is None else text
  if not text:
    return None
  cfg = json.loads(text)
  if "cluster" not in cfg:
    raise ValueError("TF_CONFIG needs a 'cluster' entry")
  cfg.setdefault("task", {"type": "worker", "index": 0})
  return cfg


def configure_cluster(worker_hosts: Optional[str] = None, task_index: int = -1,
                      ps_hosts: Optional[str] = None, task_type: Optional[str] = None) -> int:
  """Sets `TF_CONFIG` from flags when needed and returns the number of workers.

  `ps_hosts` / `task_type` extend the reference's two flags so that a parameter-server
  cluster (`{'cluster': {'ps': [...], 'worker': [...]}, 'task': {...}}`, the literal at the top
  of `ps_server/resnet_imagenet_main_dist_ps_0.py:40-50`) can be given on the command line.

  Behaviour of the reference's `distribution_utils.configure_cluster`
  (call site `resnet_imagenet_main.py:109-110`): an existing `TF_CONFIG` wins and
  the count is chief + worker tasks; otherwise `--worker_hosts` builds a
  worker-only cluster (`--task_index` is mandatory when there is more than one
  host); otherwise one worker.
  """
  cfg = parse_tf_config()
  if cfg is not None:
    spec = ClusterSpec({k: list(v) for k, v in cfg["cluster"].items()})
    return max(spec.num_workers, 1)
  if worker_hosts:
    workers = [h.strip() for h in worker_hosts.split(",") if h.strip()]
    n = len(workers)
    if n > 1 and task_index < 0:
      raise ValueError("Must specify task_index when number of workers > 1")
    task_index = 0 if n == 1 and task_index < 0 else task_index
    cluster = {"worker": workers}
    if ps_hosts:
      cluster = {"ps": [h.strip() for h in ps_hosts.split(",") if h.strip()], "worker": workers}
    task_type = task_type or "worker"
    if task_type not in cluster or not 0 <= task_index < len(cluster[task_type]):
      raise ValueError("task %s/%d is not part of the cluster %s" % (task_type, task_index, cluster))
    os.environ["TF_CONFIG"] = json.dumps({
        "cluster": cluster, "task": {"type": task_type, "index": task_index}})
    return n
  return 1


def _split_hostport(hp: str) -> Tuple[str, int]:
  host, _, port = hp.rpartition(":")
  return (host or hp), int(port) if port else 29500


def rendezvous_endpoint(spec: ClusterSpec) -> Tuple[str, int]:
  """Address every rank dials for the store: the first task in global order.

  The port is offset so the store does not collide with a TF-style service
  that may be documented on the task port itself.
  """
  job, idx = spec.ordered_tasks()[0]
  host, port = _split_hostport(spec.task_address(job, idx))
  if host in ("localhost",):
    host = "127.0.0.1"
  return host, port


def resolve_task(gpus_per_task: int = 1, local_rank: Optional[int] = None) -> TaskInfo:
  """Builds `TaskInfo` from launcher env / `TF_CONFIG`.

  `gpus_per_task` is `--num_gpus` for the synchronous strategies (each TF
  "worker" fans out into that many ranks) and 1 for parameter-server jobs,
  where the reference also runs one GPU per worker process
  (`README.md:181-194`).
  """
  env = os.environ
  cfg = parse_tf_config()
  spec = None
  if cfg is not None:
    spec = ClusterSpec({k: list(v) for k, v in cfg["cluster"].items()})

  if "RANK" in env and "WORLD_SIZE" in env:
    # Launched one-process-per-GPU: the environment is authoritative.
    rank = int(env["RANK"])
    world = int(env["WORLD_SIZE"])
    num_ps = int(env.get("DTF_NUM_PS", "0"))
    job = "ps" if rank < num_ps else "worker"
    info = TaskInfo(
        job=job, task_index=rank if job == "ps" else rank - num_ps, rank=rank,
        local_rank=int(env.get("LOCAL_RANK", rank)), world_size=world,
        num_workers=world - num_ps, num_ps=num_ps,
        master_addr=env.get("MASTER_ADDR", "127.0.0.1"),
        master_port=int(env.get("MASTER_PORT", "29500")), cluster=spec)
    return info

  if spec is None:
    return TaskInfo()

  tasks = spec.ordered_tasks()
  job = cfg["task"].get("type", "worker")
  index = int(cfg["task"].get("index", 0))
  if (job, index) not in tasks:
    raise ValueError("TF_CONFIG task %s/%d is not part of the cluster %s"
                     % (job, index, spec.as_dict()))
  g = max(int(gpus_per_task), 1)
  lr = 0 if local_rank is None else int(local_rank)
  num_ps = spec.num_ps
  # PS tasks are always single-rank; compute tasks fan out into g ranks each.
  if job == "ps":
    rank = index
  else:
    compute_tasks = [t for t in tasks if t[0] != "ps"]
    rank = num_ps + compute_tasks.index((job, index)) * g + lr
  world = num_ps + spec.num_workers * g
  host, port = rendezvous_endpoint(spec)
  return TaskInfo(job=job, task_index=index, rank=rank, local_rank=lr,
                  world_size=world, num_workers=spec.num_workers * g, num_ps=num_ps,
                  master_addr=host, master_port=port, cluster=spec)
