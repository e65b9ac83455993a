"""Small driver helpers (`official.utils.misc.model_helpers` surface)."""
from __future__ import annotations

import os
import shutil

from . import logging


def apply_clean(flags_obj):
  """`--clean`: remove `model_dir` if it exists (reference call `resnet_imagenet_main.py:275`)."""
  if flags_obj.clean and flags_obj.model_dir and os.path.isdir(flags_obj.model_dir):
    logging.info("--clean flag set. Removing existing model dir: %s", flags_obj.model_dir)
    shutil.rmtree(flags_obj.model_dir)


def past_stop_threshold(stop_threshold, eval_metric):
  """True when `eval_metric >= stop_threshold` (None never stops)."""
  if stop_threshold is None:
    return False
  if not isinstance(stop_threshold, (int, float)):
    raise ValueError("Threshold for checking stop conditions must be a number.")
  if not isinstance(eval_metric, (int, float)):
    raise ValueError("Eval metric being checked against stop conditions must be a number.")
  if eval_metric >= stop_threshold:
    logging.info("Stop threshold of %s was passed with metric value %s.",
                 stop_threshold, eval_metric)
    return True
  return False
