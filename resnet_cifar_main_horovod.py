"""ResNet-56 / CIFAR-10 with the Horovod-style API (one process per GPU).

Counterpart of `image_classification/resnet_cifar_main_horovod.py`: learning rate
scaled by `hvd.size()` (`:164`), every rank runs a full epoch of steps (`:244-245`).

    python -m distributed_tensorflow_amd.launch -np 8 resnet_cifar_main_horovod.py \
        --distribution_strategy off --use_synthetic_data
"""
from __future__ import annotations

import resnet_cifar_main as base
from distributed_tensorflow_amd.train import driver
from distributed_tensorflow_amd.utils import benchmark_logger as logger
from distributed_tensorflow_amd.utils import flags as flags_core
from distributed_tensorflow_amd.utils import logging

FLAGS = flags_core.FLAGS
LR_SCHEDULE = base.LR_SCHEDULE
learning_rate_schedule = base.learning_rate_schedule
define_cifar_flags = base.define_cifar_flags


def run(flags_obj):
  return driver.run_classifier(flags_obj, base.dataset_config(), use_horovod=True)


def main(_):
  with logger.benchmark_context(FLAGS):
    stats = run(FLAGS)
  logging.info("Run stats:\n%s", stats)
  return 0


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_cifar_flags()
  flags_core.run(main)
