"""ResNet-50 / ImageNet, asynchronous parameter server: one ps task + N worker tasks.

Replaces the sixteen files `ps_server/resnet_imagenet_main_dist_ps_{0..15}.py` (file 0 is the
ps task, file k >= 1 is worker k-1, each with the same 1 ps + 15 workers `TF_CONFIG` literal
and `CUDA_VISIBLE_DEVICES = k mod 4`: `resnet_imagenet_main_dist_ps_0.py:38-50`,
`..._ps_4.py:39-51`).  `--proc_index k` reproduces that numbering: k = 0 is the ps task, k >= 1
is worker k-1 and the process uses GPU `k mod gpus_per_host`.  Each worker trains
`train_steps // num_workers` steps per epoch (`..._ps_0.py:263`).

    python ps_server/resnet_imagenet_main_dist_ps.py --proc_index 0 --ps_hosts h0:1111 \\
        --worker_hosts h0:1112,h0:1113,... --batch_size 192 --train_epochs 1 --use_synthetic_data

`ps_server/run.sh` starts all processes of a job; on a single node
`python resnet_imagenet_main.py --distribution_strategy parameter_server --num_gpus 8` does the
same through the built-in launcher.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")

import resnet_imagenet_main as base  # noqa: E402  pylint: disable=wrong-import-position
from distributed_tensorflow_amd.utils import flags as flags_core  # noqa: E402
from distributed_tensorflow_amd.utils import logging  # noqa: E402

run = base.run
main = base.main


def _apply_proc_index(argv):
  """`--proc_index k` -> (`--task_type`, `--task_index`, visible GPU), the reference's file numbering."""
  out, k = [], None
  it = iter(argv)
  for a in it:
    if a == "--proc_index":
      k = int(next(it))
    elif a.startswith("--proc_index="):
      k = int(a.split("=", 1)[1])
    else:
      out.append(a)
  if k is not None:
    out += ["--task_type", "ps" if k == 0 else "worker", "--task_index", str(max(k - 1, 0))]
    per_host = int(os.environ.get("DTF_GPUS_PER_HOST", "8"))
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(k % per_host))
  return out


def define_flags():
  base.define_imagenet_keras_flags()
  flags_core.set_defaults(distribution_strategy="parameter_server", ps_hosts="localhost:1111",
                          worker_hosts="localhost:1112", task_type="worker", task_index=0)


def run_numbered(k: int, num_workers: int = 15):
  """Entry point of the per-process files `resnet_imagenet_main_dist_ps_<k>.py`: process k of a
  1 ps + `num_workers` workers job whose addresses come from `DTF_PS_HOSTS` /
  `DTF_WORKER_HOSTS` (default: consecutive ports from 1111 on this machine)."""
  logging.set_verbosity(logging.INFO)
  define_flags()
  ps = os.environ.get("DTF_PS_HOSTS", "localhost:1111")
  workers = os.environ.get("DTF_WORKER_HOSTS") or ",".join(
      "localhost:%d" % (1112 + i) for i in range(num_workers))
  os.environ.setdefault("DTF_GPUS_PER_HOST", "4")        # the reference pins GPU k mod 4
  sys.argv = _apply_proc_index(
      [sys.argv[0], "--proc_index", str(k), "--ps_hosts", ps, "--worker_hosts", workers]
      + sys.argv[1:])
  flags_core.run(main)


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_flags()
  sys.argv = _apply_proc_index(sys.argv)
  flags_core.run(main)
