"""`horovodrun`-style command line on top of the bundled launcher.

The reference starts its Horovod jobs with

    horovodrun --start-timeout 60 -np 16 -H hostA:4,hostB:4,hostC:4,hostD:4 --verbose \\
        python resnet_imagenet_main_horovod.py --data_dir ... --distribution_strategy off

(`README.md:8-9,239-250`): `-np` ranks in total, `-H host:slots,...` says how many of them run
on which host, the same file path must exist on every host.  This module accepts that command
line:

    python -m distributed_tensorflow_amd.horovodrun -np 8 -H localhost:8 python resnet_imagenet_main_horovod.py ...

Ranks are numbered host-major in the order of `-H` (Horovod's default), one process per slot,
`LOCAL_RANK` = slot.  The first host runs the rendezvous store.  Hosts other than this machine
are reached with `ssh host <launcher command>` (like the reference's `run.sh`, password-less
ssh and identical paths are assumed); every host runs `distributed_tensorflow_amd.launch` for
its slots, so a failing rank tears down its node's group and the job ends instead of hanging.
"""
from __future__ import annotations

import argparse
import os
import shlex
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

from . import launch as launch_lib

__all__ = ["parse_hosts", "plan", "main"]

_LOCAL = ("localhost", "127.0.0.1", "::1")


def parse_hosts(spec: Optional[str], np_total: int) -> List[Tuple[str, int]]:
  """`hostA:4,hostB:4` -> [(hostA, 4), (hostB, 4)]; no `-H`: all ranks on this machine."""
  if not spec:
    return [("localhost", np_total)]
  if " " in spec:
    raise ValueError("-H takes host:slots,host:slots without spaces")
  out = []
  for item in spec.split(","):
    host, _, slots = item.rpartition(":")
    if not host:
      host, slots = item, "1"
    out.append((host, int(slots)))
  if sum(s for _, s in out) < np_total:
    raise ValueError("-np %d exceeds the %d slots of -H %s" % (np_total, sum(s for _, s in out), spec))
  return out


def plan(hosts: List[Tuple[str, int]], np_total: int) -> List[Tuple[str, int, int]]:
  """(host, first rank, ranks on it) for the hosts that get at least one rank."""
  out, rank = [], 0
  for host, slots in hosts:
    n = min(slots, np_total - rank)
    if n <= 0:
      break
    out.append((host, rank, n))
    rank += n
  return out


def _is_local(host: str) -> bool:
  return host in _LOCAL or host == socket.gethostname()


def main(argv: Optional[Sequence[str]] = None) -> int:
  ap = argparse.ArgumentParser(prog="horovodrun", description=__doc__,
                               formatter_class=argparse.RawTextHelpFormatter)
  ap.add_argument("-np", "--num-proc", dest="np", type=int, required=True)
  ap.add_argument("-H", "--hosts", dest="hosts", default=None)
  ap.add_argument("--start-timeout", type=int, default=30, help="accepted for compatibility")
  ap.add_argument("--verbose", action="store_true")
  ap.add_argument("--master-port", type=int, default=None)
  ap.add_argument("--log-dir", default=None, help="write log<rank>.log per rank")
  ap.add_argument("command", nargs=argparse.REMAINDER, help="python script.py [args]")
  a = ap.parse_args(argv)
  cmd = list(a.command)
  if cmd and cmd[0] == "--":
    cmd = cmd[1:]
  if cmd and os.path.basename(cmd[0]).startswith("python"):
    cmd = cmd[1:]                  # the launcher starts the interpreter itself
  if not cmd:
    ap.error("no command given")
  nodes = plan(parse_hosts(a.hosts, a.np), a.np)
  if len(nodes) == 1 and _is_local(nodes[0][0]):
    if a.verbose:
      print("horovodrun: %d ranks on this machine" % a.np, file=sys.stderr)
    return launch_lib.launch(cmd, a.np, master_port=a.master_port, log_dir=a.log_dir)
  # several hosts: one launcher per host, node-major ranks need equal slots per used host
  sizes = {n for _, _, n in nodes}
  if len(sizes) != 1:
    raise ValueError("every host must run the same number of ranks (got %s)" % sorted(sizes))
  per = sizes.pop()
  master = nodes[0][0]
  master_addr = "127.0.0.1" if _is_local(master) and len(nodes) == 1 else master
  port = a.master_port or launch_lib.free_port()
  procs = []
  for i, (host, _first, _n) in enumerate(nodes):
    node_cmd = [sys.executable, "-m", "distributed_tensorflow_amd.launch", "--nproc", str(per),
                "--nnodes", str(len(nodes)), "--node_rank", str(i), "--master_addr", master_addr,
                "--master_port", str(port)]
    if a.log_dir:
      node_cmd += ["--log_dir", a.log_dir]
    node_cmd += cmd
    if _is_local(host):
      full = node_cmd
    else:
      remote = "cd %s && %s" % (shlex.quote(os.getcwd()), " ".join(shlex.quote(c) for c in node_cmd))
      full = ["ssh", "-o", "ConnectTimeout=%d" % a.start_timeout, host, remote]
    if a.verbose:
      print("horovodrun: node %d on %s: %s" % (i, host, " ".join(full)), file=sys.stderr)
    procs.append(subprocess.Popen(full))
  code = 0
  try:
    for p in procs:
      rc = p.wait()
      if rc != 0 and code == 0:
        code = rc
        for q in procs:            # one node failed: stop the others
          if q.poll() is None:
            q.terminate()
  except KeyboardInterrupt:
    for q in procs:
      if q.poll() is None:
        q.terminate()
    code = 130
  return code


if __name__ == "__main__":
  sys.exit(main())
