"""Flags, cluster-spec parsing and strategy selection (pure Python, CPU)."""
import json
import os

import pytest

from distributed_tensorflow_amd import common
from distributed_tensorflow_amd.parallel import cluster, strategy
from distributed_tensorflow_amd.utils import flags


def _define():
  common.define_keras_flags()
  return flags.FLAGS


def test_absl_syntax_and_aliases():
  F = _define()
  rest = F(["prog", "-distribution_strategy", "off", "--batch_size=4", "-bs", "8",
            "-use_synthetic_data", "true", "--noskip_eval", "--dtype", "bf16",
            "--num_gpus=0", "-ds", "mirrored", "positional"])
  assert F.distribution_strategy == "mirrored"
  assert F.batch_size == 8
  assert F.use_synthetic_data is True
  assert F.skip_eval is False
  assert F.dtype == "bf16"
  # a boolean never swallows the next token: 'true' stays positional
  assert rest[1:] == ["true", "positional"]


def test_defaults_and_set_defaults():
  F = _define()
  flags.set_defaults(train_epochs=182, batch_size=128)
  F(["prog"])
  assert F.train_epochs == 182 and F.batch_size == 128
  assert F.log_steps == 100 and F.num_packs == 1 and F.task_index == -1
  assert F.distribution_strategy == "default"
  assert F.batchnorm_spatial_persistent is True and F.report_accuracy_metrics is True


def test_unknown_flag_and_bad_value():
  F = _define()
  with pytest.raises(flags.UnrecognizedFlagError):
    F(["prog", "--no_such_flag=1"])
  with pytest.raises(flags.IllegalFlagValueError):
    F(["prog", "--batch_size", "abc"])
  with pytest.raises(flags.IllegalFlagValueError):
    F(["prog", "--dtype", "int8"])


def test_loss_scale_and_dtype():
  F = _define()
  F(["prog", "--dtype", "fp16"])
  assert flags.get_loss_scale(F, default_for_fp16=128) == 128
  assert flags.get_dtype(F) == "float16"
  F(["prog", "--dtype", "fp16", "--loss_scale", "dynamic"])
  assert flags.get_loss_scale(F) == "dynamic"
  F(["prog", "--dtype", "fp32", "--loss_scale", "8"])
  assert flags.get_loss_scale(F) == 8.0
  F.unparse_flags()
  F(["prog", "--dtype", "bf16"])
  assert flags.get_loss_scale(F) == 1


def test_configure_cluster_from_worker_hosts(monkeypatch):
  monkeypatch.delenv("TF_CONFIG", raising=False)
  assert cluster.configure_cluster(None, -1) == 1
  with pytest.raises(ValueError):
    cluster.configure_cluster("a:1,b:2", -1)
  assert cluster.configure_cluster("a:1,b:2", 1) == 2
  cfg = json.loads(os.environ["TF_CONFIG"])
  assert cfg == {"cluster": {"worker": ["a:1", "b:2"]}, "task": {"type": "worker", "index": 1}}
  monkeypatch.delenv("TF_CONFIG", raising=False)


def test_tf_config_roles(monkeypatch):
  """The 1 PS + 15 workers job of the reference: PS is rank 0, worker k is rank k+1."""
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    monkeypatch.delenv(k, raising=False)
  hosts = ["h%d:%d" % (i // 4, 1111 + i % 4) for i in range(16)]
  spec = {"ps": hosts[:1], "worker": hosts[1:]}
  monkeypatch.setenv("TF_CONFIG", json.dumps({"cluster": spec, "task": {"type": "ps", "index": 0}}))
  t = cluster.resolve_task()
  assert (t.job, t.rank, t.world_size, t.num_workers, t.num_ps) == ("ps", 0, 16, 15, 1)
  assert not t.is_chief
  monkeypatch.setenv("TF_CONFIG", json.dumps({"cluster": spec,
                                              "task": {"type": "worker", "index": 14}}))
  t = cluster.resolve_task()
  assert (t.job, t.rank, t.worker_rank) == ("worker", 15, 14)
  assert cluster.configure_cluster() == 15
  monkeypatch.setenv("TF_CONFIG", json.dumps({"cluster": spec,
                                              "task": {"type": "worker", "index": 0}}))
  assert cluster.resolve_task().is_chief
  assert cluster.resolve_task().master_addr == "h0"


def test_multi_worker_ranks_fan_out(monkeypatch):
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    monkeypatch.delenv(k, raising=False)
  spec = {"worker": ["10.0.0.1:8080", "10.0.0.2:8080"]}
  monkeypatch.setenv("TF_CONFIG", json.dumps({"cluster": spec,
                                              "task": {"type": "worker", "index": 1}}))
  t = cluster.resolve_task(gpus_per_task=8, local_rank=3)
  assert (t.rank, t.world_size, t.num_workers) == (11, 16, 16)
  assert (t.master_addr, t.master_port) == ("10.0.0.1", 8080)


def test_launcher_env_wins(monkeypatch):
  monkeypatch.setenv("RANK", "3")
  monkeypatch.setenv("WORLD_SIZE", "8")
  monkeypatch.setenv("LOCAL_RANK", "3")
  monkeypatch.setenv("DTF_NUM_PS", "1")
  monkeypatch.setenv("MASTER_PORT", "12345")
  t = cluster.resolve_task()
  assert (t.job, t.rank, t.worker_rank, t.num_workers, t.master_port) == ("worker", 3, 2, 7, 12345)


def test_strategy_selection(monkeypatch):
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    monkeypatch.delenv(k, raising=False)
  assert strategy.get_distribution_strategy("off", 0) is None
  assert strategy.get_distribution_strategy("OFF", 1) is None
  with pytest.raises(ValueError):
    strategy.get_distribution_strategy("off", 2)
  s = strategy.get_distribution_strategy("default", 0)
  assert isinstance(s, strategy.OneDeviceStrategy) and s.num_replicas_in_sync == 1
  with pytest.raises(ValueError):
    strategy.get_distribution_strategy("one_device", 2)
  with pytest.raises(ValueError):
    strategy.get_distribution_strategy("bogus", 0)
  assert strategy.strategy_needs_launch("mirrored", 8) == 8
  assert strategy.strategy_needs_launch("off", 1) == 0
  monkeypatch.setenv("RANK", "0")
  assert strategy.strategy_needs_launch("mirrored", 8) == 0
