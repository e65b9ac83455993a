"""Contracts of build_stats / TimeHistory / schedules (reference `common_test.py`)."""
from unittest import mock

import numpy as np
import pytest

from distributed_tensorflow_amd import common
from distributed_tensorflow_amd.train import schedules
from distributed_tensorflow_amd.train.callbacks import (BatchTimestamp, TimeHistory,
                                                        get_profiler_callback)


def _build_history(loss, cat_accuracy=None, cat_accuracy_sparse=None):
  history_p = mock.Mock()
  history = {"loss": [np.float64(loss)]}
  if cat_accuracy:
    history["categorical_accuracy"] = [np.float64(cat_accuracy)]
  if cat_accuracy_sparse:
    history["sparse_categorical_accuracy"] = [np.float64(cat_accuracy_sparse)]
  history_p.history = history
  return history_p


def _build_eval_output(top_1, eval_loss):
  return [np.float64(eval_loss), np.float64(top_1)]


def test_build_stats():
  history = _build_history(1.145, cat_accuracy=.99988)
  eval_output = _build_eval_output(.56, 5.990)
  th = TimeHistory(128, 100)
  th.timestamp_log = [BatchTimestamp(0, 1), BatchTimestamp(1, 2), BatchTimestamp(2, 3)]
  th.train_finish_time = 12345
  stats = common.build_stats(history, eval_output, [th])
  assert stats["loss"] == 1.145
  assert stats["training_accuracy_top_1"] == .99988
  assert stats["accuracy_top_1"] == .56
  assert stats["eval_loss"] == 5.990
  assert stats["step_timestamp_log"] == th.timestamp_log
  assert stats["train_finish_time"] == 12345
  # batch_size * log_steps * (len - 1) / (t_last - t_first)
  assert stats["avg_exp_per_second"] == 128 * 100 * 2 / 2


def test_build_stats_sparse():
  history = _build_history(1.145, cat_accuracy_sparse=.99988)
  stats = common.build_stats(history, _build_eval_output(.928, 1.9844), None)
  assert stats["loss"] == 1.145
  assert stats["training_accuracy_top_1"] == .99988
  assert stats["accuracy_top_1"] == .928
  assert stats["eval_loss"] == 1.9844
  assert "step_timestamp_log" not in stats


def test_time_history():
  th = TimeHistory(batch_size=128, log_steps=3)
  th.on_train_begin()
  th.on_epoch_begin(0)
  for i in range(7):
    th.on_batch_begin(i)
    th.on_batch_end(i)
  th.on_epoch_end(0)
  th.on_train_end()
  # first step + steps 3 and 6
  assert len(th.timestamp_log) == 3
  assert [t.batch_index for t in th.timestamp_log] == [1, 3, 6]
  assert th.train_finish_time is not None


def test_imagenet_schedule():
  f = schedules.imagenet_learning_rate_schedule
  bpe = 5004.0
  assert f(0, 0, bpe, 256) == 0.0
  assert f(2, 2502, bpe, 256) == pytest.approx(0.1 * 2.5 / 5)
  assert f(5, 0, bpe, 256) == pytest.approx(0.1)
  assert f(29, 5003, bpe, 512) == pytest.approx(0.2)
  assert f(30, 0, bpe, 256) == pytest.approx(0.01)
  assert f(60, 0, bpe, 256) == pytest.approx(0.001)
  assert f(85, 0, bpe, 256) == pytest.approx(0.0001)


def test_cifar_schedule():
  f = schedules.cifar_learning_rate_schedule
  assert f(0, 0, 390, 128) == pytest.approx(0.1)
  assert f(90, 0, 390, 256) == pytest.approx(0.2)
  assert f(91, 0, 390, 128) == pytest.approx(0.01)
  assert f(136, 5, 390, 128) == pytest.approx(0.001)
  assert f(182, 0, 390, 128) == pytest.approx(0.0001)


def test_piecewise_constant_decay_with_warmup():
  s = schedules.PiecewiseConstantDecayWithWarmup(
      batch_size=256, epoch_size=1281167, warmup_epochs=5, boundaries=[30, 60, 80],
      multipliers=[1.0, 0.1, 0.01, 0.001])
  spe = 1281167 // 256
  assert s(0) == 0.0
  assert s(spe * 5 // 2) == pytest.approx(0.1 * (spe * 5 // 2) / (spe * 5))
  assert s(spe * 5) == pytest.approx(0.1)
  assert s(spe * 30) == pytest.approx(0.1)          # boundary belongs to the left piece
  assert s(spe * 30 + 1) == pytest.approx(0.01)
  assert s(spe * 80 + 1) == pytest.approx(0.0001)
  with pytest.raises(ValueError):
    schedules.PiecewiseConstantDecayWithWarmup(256, 1000, 5, [1, 2], [1.0, 0.1])


def test_profile_steps_validation():
  cb = get_profiler_callback("/tmp/x", "2,4")
  assert (cb.start_step, cb.stop_step) == (2, 4)
  for bad in ("a,b", "3", "5,2", "-1,2"):
    with pytest.raises(ValueError):
      get_profiler_callback("/tmp/x", bad)
