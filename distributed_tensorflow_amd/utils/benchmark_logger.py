"""Benchmark run logging (`official.utils.logs.logger` surface).

`benchmark_context(flags)` wraps a run: it records run information (flags, device,
framework versions) at entry and the final status (`success` / `failure`) at exit,
either to the process log (`BaseBenchmarkLogger`) or as JSON lines under
`--benchmark_log_dir` (`BenchmarkFileLogger`).  Reference call sites:
`resnet_cifar_main.py:234`, `resnet_imagenet_main.py:276`.
"""
from __future__ import annotations

import contextlib
import datetime
import json
import os
import threading

from . import logging

METRIC_LOG_FILE_NAME = "metric.log"
BENCHMARK_RUN_LOG_FILE_NAME = "benchmark_run.log"
RUN_STATUS_SUCCESS = "success"
RUN_STATUS_FAILURE = "failure"
RUN_STATUS_RUNNING = "running"

_logger = None
_lock = threading.Lock()


def _now():
  return datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%S.%fZ")


class BaseBenchmarkLogger:
  """Logs metrics and run info to the process log."""

  def log_metric(self, name, value, unit=None, global_step=None, extras=None):
    try:
      value = float(value)
    except (TypeError, ValueError):
      logging.warning("Metric value to log should be a number. Got %s", type(value))
      return
    logging.info("Benchmark metric: %s", dict(
        name=name, value=value, unit=unit, global_step=global_step, timestamp=_now(),
        extras=extras))

  def log_evaluation_result(self, eval_results):
    step = eval_results.get("global_step", 0)
    for k in sorted(eval_results):
      if k != "global_step":
        self.log_metric(k, eval_results[k], global_step=step)

  def log_run_info(self, model_name, dataset_name, run_params, test_id=None):
    logging.info("Benchmark run: %s", _gather_run_info(model_name, dataset_name, run_params,
                                                       test_id))

  def on_finish(self, status):
    logging.info("Benchmark run status: %s", status)


class BenchmarkFileLogger(BaseBenchmarkLogger):
  """Writes metrics / run info as JSON lines under `logging_dir`."""

  def __init__(self, logging_dir):
    self._dir = logging_dir
    os.makedirs(logging_dir, exist_ok=True)
    self._metric_f = open(os.path.join(logging_dir, METRIC_LOG_FILE_NAME), "a")

  def log_metric(self, name, value, unit=None, global_step=None, extras=None):
    try:
      value = float(value)
    except (TypeError, ValueError):
      return
    json.dump(dict(name=name, value=value, unit=unit, global_step=global_step,
                   timestamp=_now(), extras=extras), self._metric_f)
    self._metric_f.write("\n")
    self._metric_f.flush()

  def log_run_info(self, model_name, dataset_name, run_params, test_id=None):
    with open(os.path.join(self._dir, BENCHMARK_RUN_LOG_FILE_NAME), "w") as f:
      json.dump(_gather_run_info(model_name, dataset_name, run_params, test_id), f)
      f.write("\n")

  def on_finish(self, status):
    self._metric_f.flush()
    self._metric_f.close()
    with open(os.path.join(self._dir, "run_status.log"), "w") as f:
      json.dump({"status": status, "timestamp": _now()}, f)


def _gather_run_info(model_name, dataset_name, run_params, test_id):
  info = {"model_name": model_name, "dataset": {"name": dataset_name},
          "run_date": _now(), "test_id": test_id, "machine_config": {}}
  try:
    import torch  # pylint: disable=import-outside-toplevel
    info["framework"] = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    info["machine_config"]["gpu_count"] = torch.cuda.device_count()
    if torch.cuda.is_available():
      info["machine_config"]["gpu_model"] = torch.cuda.get_device_name(0)
    info["machine_config"]["cpu_cores"] = os.cpu_count()
  except Exception:  # pragma: no cover
    pass
  if run_params:
    info["run_parameters"] = [{"name": k, "value": repr(v)} for k, v in sorted(run_params.items())]
  return info


def config_benchmark_logger(flag_obj=None):
  global _logger
  with _lock:
    kind = getattr(flag_obj, "benchmark_logger_type", "BaseBenchmarkLogger") if flag_obj else \
        "BaseBenchmarkLogger"
    if kind == "BenchmarkFileLogger" and getattr(flag_obj, "benchmark_log_dir", None):
      _logger = BenchmarkFileLogger(flag_obj.benchmark_log_dir)
    else:
      _logger = BaseBenchmarkLogger()
  return _logger


def get_benchmark_logger():
  if _logger is None:
    config_benchmark_logger()
  return _logger


@contextlib.contextmanager
def benchmark_context(flag_obj):
  """Context of benchmark, which will update status of the run accordingly."""
  blog = config_benchmark_logger(flag_obj)
  try:
    yield
    blog.on_finish(RUN_STATUS_SUCCESS)
  except Exception:  # pylint: disable=broad-except
    blog.on_finish(RUN_STATUS_FAILURE)
    raise
