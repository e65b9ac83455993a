"""ResNet-56 / CIFAR-10 under `multi_worker_mirrored`: worker 1 of the localhost pair.

Counterpart of `image_classification/resnet_cifar_main_dist_1.py:35-42` (workers
`localhost:8080` / `localhost:8081`, this process is `worker/1`); start it next to
`resnet_cifar_main_dist.py` on the same machine to rehearse a two-worker job:

    python resnet_cifar_main_dist.py   --use_synthetic_data --num_gpus 0 --train_steps 5 &
    python resnet_cifar_main_dist_1.py --use_synthetic_data --num_gpus 0 --train_steps 5
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
  flags_core.set_defaults(distribution_strategy="multi_worker_mirrored",
                          worker_hosts="localhost:8080,localhost:8081", task_index=1)


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_flags()
  flags_core.run(main)
