"""ResNet-50 / ImageNet with the Horovod-style API (one process per GPU).

Counterpart of `image_classification/resnet_imagenet_main_horovod.py`:
`--distribution_strategy off`, gradients averaged by `hvd.DistributedOptimizer`
(fused RCCL all-reduce), rank-0 broadcast, metric averaging, LR warm-up, rank-0
checkpoints `./checkpoint-{epoch}.h5`, `steps_per_epoch = train_steps // size`.

    python -m distributed_tensorflow_amd.launch -np 8 resnet_imagenet_main_horovod.py \
        --distribution_strategy off --use_synthetic_data --dtype bf16 --batch_size 256
"""
from __future__ import annotations

import resnet_imagenet_main as base
from distributed_tensorflow_amd.train import driver
from distributed_tensorflow_amd.utils import benchmark_logger as logger
from distributed_tensorflow_amd.utils import flags as flags_core
from distributed_tensorflow_amd.utils import logging, model_helpers

FLAGS = flags_core.FLAGS
LR_SCHEDULE = base.LR_SCHEDULE
learning_rate_schedule = base.learning_rate_schedule
define_imagenet_keras_flags = base.define_imagenet_keras_flags


def run(flags_obj):
  return driver.run_classifier(flags_obj, base.dataset_config(), use_horovod=True)


def main(_):
  model_helpers.apply_clean(FLAGS)
  with logger.benchmark_context(FLAGS):
    stats = run(FLAGS)
  logging.info("Run stats:\n%s", stats)
  return 0


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_imagenet_keras_flags()
  flags_core.run(main)
