"""`build_stats`: normalises a training run into the reference's result dictionary.

Keys (reference `common.py:202-245`; a real sample is `ps_server/log1.log:163`):
`loss`, `training_accuracy_top_1`, `accuracy_top_1`, `eval_loss`,
`step_timestamp_log`, `train_finish_time`, `avg_exp_per_second`.
"""
from __future__ import annotations

TRAIN_TOP_1 = "training_accuracy_top_1"


def _item(x):
  return x.item() if hasattr(x, "item") else x


def build_stats(history, eval_output, callbacks):
  from .callbacks import TimeHistory  # pylint: disable=import-outside-toplevel
  stats = {}
  if eval_output:
    stats["accuracy_top_1"] = _item(eval_output[1])
    stats["eval_loss"] = _item(eval_output[0])

  if history and history.history:
    train_hist = history.history
    stats["loss"] = _item(train_hist["loss"][-1])
    if "categorical_accuracy" in train_hist:
      stats[TRAIN_TOP_1] = _item(train_hist["categorical_accuracy"][-1])
    elif "sparse_categorical_accuracy" in train_hist:
      stats[TRAIN_TOP_1] = _item(train_hist["sparse_categorical_accuracy"][-1])

  if not callbacks:
    return stats

  for callback in callbacks:
    if isinstance(callback, TimeHistory):
      timestamp_log = callback.timestamp_log
      stats["step_timestamp_log"] = timestamp_log
      stats["train_finish_time"] = callback.train_finish_time
      if len(timestamp_log) > 1:
        stats["avg_exp_per_second"] = (
            callback.batch_size * callback.log_steps * (len(callback.timestamp_log) - 1) /
            (timestamp_log[-1].timestamp - timestamp_log[0].timestamp))
  return stats
