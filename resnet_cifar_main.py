"""Runs a ResNet-56 model on the CIFAR-10 dataset (MI355X-native engine).

Entry point with the command line of the reference driver
(`image_classification/resnet_cifar_main.py`).

    python resnet_cifar_main.py --use_synthetic_data --distribution_strategy off \
        --num_gpus 0 --batch_size 128 --train_steps 10
"""
from __future__ import annotations

import sys

from distributed_tensorflow_amd import common
from distributed_tensorflow_amd.data import cifar_preprocessing
from distributed_tensorflow_amd.models import resnet_cifar_model
from distributed_tensorflow_amd.train import driver
from distributed_tensorflow_amd.train.schedules import (CIFAR_LR_SCHEDULE as LR_SCHEDULE,
                                                        cifar_learning_rate_schedule)
from distributed_tensorflow_amd.utils import benchmark_logger as logger
from distributed_tensorflow_amd.utils import flags as flags_core
from distributed_tensorflow_amd.utils import logging

FLAGS = flags_core.FLAGS
learning_rate_schedule = cifar_learning_rate_schedule


def dataset_config() -> driver.DatasetConfig:
  return driver.DatasetConfig(
      name="cifar10",
      height=cifar_preprocessing.HEIGHT, width=cifar_preprocessing.WIDTH,
      channels=cifar_preprocessing.NUM_CHANNELS,
      num_classes=cifar_preprocessing.NUM_CLASSES,
      num_images=cifar_preprocessing.NUM_IMAGES,
      model_fn=lambda: resnet_cifar_model.resnet56(classes=cifar_preprocessing.NUM_CLASSES),
      loss="categorical_crossentropy",
      accuracy_metric="categorical_accuracy",
      learning_rate_schedule=learning_rate_schedule,
      lr_schedule_table=None,
      real_input_fn=cifar_preprocessing.input_fn,
      parse_record_fn=cifar_preprocessing.parse_record,
      allow_fp16=False,
      scale_lr_by_size_in_horovod=True,
      divide_steps_by_size_in_horovod=False)


def run(flags_obj):
  """Run ResNet CIFAR-10 training and eval loop. Returns the stats dictionary."""
  return driver.run_classifier(flags_obj, dataset_config())


def define_cifar_flags():
  common.define_keras_flags(dynamic_loss_scale=False)
  flags_core.set_defaults(data_dir="/tmp/cifar10_data/cifar-10-batches-bin",
                          model_dir="/tmp/cifar10_model",
                          train_epochs=182, epochs_between_evals=10, batch_size=128)


def main(_):
  code = driver.maybe_launch(FLAGS, sys.argv)
  if code is not None:
    return code
  with logger.benchmark_context(FLAGS):
    stats = run(FLAGS)
  logging.info("Run stats:\n%s", stats)
  return 0


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_cifar_flags()
  flags_core.run(main)
