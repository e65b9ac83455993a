"""ResNet-56 / CIFAR-10 with the asynchronous parameter-server strategy.

One file replaces `image_classification/resnet_cifar_main_dist_ps_{0,1}.py` and
`..._ps_remote_{0,1}.py` (identical drivers whose `TF_CONFIG` literal names a `ps` task on
`localhost:8080` and a `worker` task on `localhost:8081`; `_0` is the ps task, `_1` the
worker, `resnet_cifar_main_dist_ps_0.py:36-48`).  The role is a flag:

    python resnet_cifar_main_dist_ps.py --task_type ps     --task_index 0 --use_synthetic_data &
    python resnet_cifar_main_dist_ps.py --task_type worker --task_index 0 --use_synthetic_data

The ps task holds the parameter and momentum shards, applies each pushed gradient on arrival
and exits once every worker has reported that it is done (the reference's ps process never
returns and is killed by hand).
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import resnet_cifar_main as base  # noqa: E402  pylint: disable=wrong-import-position
from distributed_tensorflow_amd.utils import flags as flags_core  # noqa: E402
from distributed_tensorflow_amd.utils import logging  # noqa: E402

run = base.run
main = base.main


def define_flags():
  base.define_cifar_flags()
  flags_core.set_defaults(distribution_strategy="parameter_server", ps_hosts="localhost:8080",
                          worker_hosts="localhost:8081", task_type="worker", task_index=0)


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_flags()
  flags_core.run(main)
