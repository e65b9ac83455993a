"""One-process-per-GPU launcher (replaces `run.sh` ssh fan-out, `horovodrun`, `kill.sh`).

The reference starts its jobs with a bash loop over ssh (`ps_server/run.sh:3-29`,
one python per GPU, stderr to `log$i.log`), with `horovodrun -np N`
(`README.md:239-250`), and stops a parameter-server job with `pkill`
(`ps_server/kill.sh`).  This launcher starts `nproc` copies of a driver on the
local node with `RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`
(and `DTF_NUM_PS` for parameter-server jobs: the first `num_ps` ranks are PS
tasks), writes one log per rank when asked, and tears the whole group down as
soon as one rank fails - fail-fast instead of a hung job.

    python -m distributed_tensorflow_amd.launch --nproc 8 resnet_imagenet_main.py ...
    python -m distributed_tensorflow_amd.launch --nproc 8 --num_ps 1 resnet_imagenet_main.py \
        --distribution_strategy parameter_server ...

For multi-node jobs run the launcher once per node with `--node_rank`, `--nnodes`
and `--master_addr` (ranks are numbered node-major).
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence

__all__ = ["launch", "free_port", "main"]


def free_port() -> int:
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def launch(argv: Sequence[str], nproc: int, num_ps: int = 0, master_addr: str = "127.0.0.1",
           master_port: Optional[int] = None, log_dir: Optional[str] = None,
           nnodes: int = 1, node_rank: int = 0, env_extra: Optional[Dict[str, str]] = None,
           stagger_s: float = 0.0, timeout_s: Optional[float] = None) -> int:
  """Starts `nproc` local ranks running `python argv...`; returns the job's exit code."""
  if master_port is None:
    master_port = free_port() if nnodes == 1 else 29500
  world = nproc * nnodes
  procs: List[subprocess.Popen] = []
  logs = []
  if log_dir:
    os.makedirs(log_dir, exist_ok=True)

  # The ranks run in their own sessions (so a rank's group can be signalled as a unit); a
  # SIGTERM / SIGHUP to the launcher must therefore be forwarded or the ranks would outlive it.
  def _on_signal(signum, _frame):
    sys.stderr.write("[launch] got signal %d; stopping the job\n" % signum)
    _kill_all(procs)
    os._exit(128 + signum)

  old_handlers = {}
  for sig in (signal.SIGTERM, signal.SIGHUP):
    try:
      old_handlers[sig] = signal.signal(sig, _on_signal)
    except ValueError:      # not in the main thread
      pass
  try:
    for lr in range(nproc):
      rank = node_rank * nproc + lr
      env = dict(os.environ)
      env.update({"RANK": str(rank), "LOCAL_RANK": str(lr), "WORLD_SIZE": str(world),
                  "LOCAL_WORLD_SIZE": str(nproc), "MASTER_ADDR": master_addr,
                  "MASTER_PORT": str(master_port), "DTF_NUM_PS": str(num_ps),
                  "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
      if env_extra:
        env.update(env_extra)        # may pin LOCAL_RANK (several ranks on one GPU in tests)
      stderr = None
      if log_dir:
        f = open(os.path.join(log_dir, "log%d.log" % rank), "w")
        logs.append(f)
        stderr = f
      procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stderr=stderr,
                                    start_new_session=True))
      if stagger_s:
        time.sleep(stagger_s)   # the reference staggers process start by 1 s (run.sh)
    deadline = time.time() + timeout_s if timeout_s else None
    code = 0
    alive = set(range(len(procs)))
    while alive:
      for i in list(alive):
        rc = procs[i].poll()
        if rc is None:
          continue
        alive.discard(i)
        if rc != 0 and code == 0:
          code = rc
          sys.stderr.write("[launch] rank %d exited with code %d; stopping the job\n"
                           % (node_rank * nproc + i, rc))
          _kill_all(procs)
      if deadline and time.time() > deadline:
        sys.stderr.write("[launch] timeout after %.0f s; stopping the job\n" % timeout_s)
        _kill_all(procs)
        code = code or 124
        break
      time.sleep(0.05)
    return code
  except KeyboardInterrupt:
    _kill_all(procs)
    return 130
  finally:
    for sig, h in old_handlers.items():
      signal.signal(sig, h)
    for f in logs:
      f.close()


def _kill_all(procs: List[subprocess.Popen]):
  """Stops exactly the process groups this launcher started."""
  for p in procs:
    if p.poll() is None:
      try:
        os.killpg(p.pid, signal.SIGTERM)
      except ProcessLookupError:
        pass
  t0 = time.time()
  while time.time() - t0 < 5 and any(p.poll() is None for p in procs):
    time.sleep(0.05)
  for p in procs:
    if p.poll() is None:
      try:
        os.killpg(p.pid, signal.SIGKILL)
      except ProcessLookupError:
        pass


def main(argv: Optional[Sequence[str]] = None) -> int:
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
  ap.add_argument("--nproc", "-np", type=int, required=True, help="ranks (GPUs) on this node")
  ap.add_argument("--num_ps", type=int, default=0, help="leading ranks that act as PS tasks")
  ap.add_argument("--nnodes", type=int, default=1)
  ap.add_argument("--node_rank", type=int, default=0)
  ap.add_argument("--master_addr", default="127.0.0.1")
  ap.add_argument("--master_port", type=int, default=None)
  ap.add_argument("--log_dir", default=None, help="write log<rank>.log (stderr of each rank)")
  ap.add_argument("--stagger", type=float, default=0.0, help="seconds between rank starts")
  ap.add_argument("script", help="driver script")
  ap.add_argument("script_args", nargs=argparse.REMAINDER)
  a = ap.parse_args(argv)
  return launch([a.script] + a.script_args, a.nproc, a.num_ps, a.master_addr, a.master_port,
                a.log_dir, a.nnodes, a.node_rank, stagger_s=a.stagger)


if __name__ == "__main__":
  sys.exit(main())
