"""Runs a ResNet-50 v1.5 model on the ImageNet dataset (MI355X-native engine).

Entry point with the command line of the reference driver
(`image_classification/resnet_imagenet_main.py`): same flags, same `run()` /
`main()` / `learning_rate_schedule()` / `define_imagenet_keras_flags()` surface.

    python resnet_imagenet_main.py --use_synthetic_data --dtype bf16 --batch_size 256 \
        --distribution_strategy mirrored --num_gpus 8 --train_steps 200 --skip_eval
"""
from __future__ import annotations

import sys

from distributed_tensorflow_amd import common
from distributed_tensorflow_amd.data import imagenet_preprocessing
from distributed_tensorflow_amd.models import resnet_model
from distributed_tensorflow_amd.models.trivial_model import trivial_model as build_trivial_model
from distributed_tensorflow_amd.train import driver
from distributed_tensorflow_amd.train.schedules import (IMAGENET_LR_SCHEDULE as LR_SCHEDULE,
                                                        imagenet_learning_rate_schedule)
from distributed_tensorflow_amd.utils import benchmark_logger as logger
from distributed_tensorflow_amd.utils import flags as flags_core
from distributed_tensorflow_amd.utils import logging, model_helpers

FLAGS = flags_core.FLAGS
learning_rate_schedule = imagenet_learning_rate_schedule


def dataset_config() -> driver.DatasetConfig:
  return driver.DatasetConfig(
      name="imagenet",
      height=imagenet_preprocessing.DEFAULT_IMAGE_SIZE,
      width=imagenet_preprocessing.DEFAULT_IMAGE_SIZE,
      channels=imagenet_preprocessing.NUM_CHANNELS,
      num_classes=imagenet_preprocessing.NUM_CLASSES,
      num_images=imagenet_preprocessing.NUM_IMAGES,
      model_fn=lambda: resnet_model.resnet50(num_classes=imagenet_preprocessing.NUM_CLASSES),
      trivial_model_fn=lambda: build_trivial_model(imagenet_preprocessing.NUM_CLASSES),
      loss="sparse_categorical_crossentropy",
      accuracy_metric="sparse_categorical_accuracy",
      learning_rate_schedule=learning_rate_schedule,
      lr_schedule_table=LR_SCHEDULE,
      real_input_fn=imagenet_preprocessing.input_fn,
      parse_record_fn=imagenet_preprocessing.parse_record,
      allow_fp16=True,
      scale_lr_by_size_in_horovod=False,
      divide_steps_by_size_in_horovod=True)


def run(flags_obj):
  """Run ResNet ImageNet training and eval loop. Returns the stats dictionary."""
  return driver.run_classifier(flags_obj, dataset_config())


def define_imagenet_keras_flags():
  common.define_keras_flags()
  flags_core.set_defaults(train_epochs=90)


def main(_):
  code = driver.maybe_launch(FLAGS, sys.argv)
  if code is not None:
    return code
  model_helpers.apply_clean(FLAGS)
  with logger.benchmark_context(FLAGS):
    stats = run(FLAGS)
  logging.info("Run stats:\n%s", stats)
  return 0


if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  define_imagenet_keras_flags()
  flags_core.run(main)
