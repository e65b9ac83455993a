"""End-to-end smoke tests of the entry points (reference: resnet_cifar_test.py,
resnet_imagenet_test.py): one training step + one eval step on synthetic data."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.data import cifar_preprocessing, imagenet_preprocessing
from distributed_tensorflow_amd.utils import flags as flags_core
from distributed_tensorflow_amd.utils import testing as integration

import resnet_cifar_main
import resnet_imagenet_main

_CIFAR_FLAGS = ["-batch_size", "4", "-train_steps", "1", "-use_synthetic_data", "true",
                "-num_gpus", "0", "-data_format", "channels_last"]


@pytest.fixture
def small_validation(monkeypatch):
  monkeypatch.setitem(cifar_preprocessing.NUM_IMAGES, "validation", 4)
  monkeypatch.setitem(imagenet_preprocessing.NUM_IMAGES, "validation", 4)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("strategy", ["off", "default", "one_device"])
def test_cifar_end_to_end(tmp_path, small_validation, strategy):
  resnet_cifar_main.define_cifar_flags()
  stats = integration.run_synthetic(
      main=resnet_cifar_main.run, tmp_root=str(tmp_path),
      extra_flags=_CIFAR_FLAGS + ["-distribution_strategy", strategy, "-log_steps", "1"])
  for k in ("loss", "training_accuracy_top_1", "accuracy_top_1", "eval_loss",
            "step_timestamp_log", "train_finish_time"):
    assert k in stats, k
  assert stats["loss"] > 0


def test_cifar_skip_eval_and_tensorboard(tmp_path, small_validation):
  resnet_cifar_main.define_cifar_flags()
  flags_core.FLAGS([sys.argv[0], "--model_dir", str(tmp_path / "m"), "--train_steps", "2",
                    "--use_synthetic_data", "--num_gpus", "0", "--batch_size", "4",
                    "--distribution_strategy", "off", "--skip_eval", "--enable_tensorboard",
                    "--log_steps", "1"])
  stats = resnet_cifar_main.run(flags_core.FLAGS)
  assert "accuracy_top_1" not in stats and "avg_exp_per_second" in stats
  ev = list((tmp_path / "m" / "train").iterdir())
  assert len(ev) == 1 and ev[0].name.startswith("events.out.tfevents")
  # the event file is valid TFRecord framing with the version header first
  from distributed_tensorflow_amd.utils import tfrecord
  with open(ev[0], "rb") as f:
    recs = list(tfrecord.read_records(f, verify=True))
  assert b"brain.Event:2" in recs[0] and any(b"epoch_loss" in r for r in recs)


def test_imagenet_trivial_model_end_to_end(tmp_path, small_validation):
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "2", "-train_steps", "1", "-num_gpus", "0",
                   "-distribution_strategy", "off", "-use_trivial_model", "-use_tensor_lr"])
  assert "loss" in stats and "accuracy_top_1" in stats


@pytest.mark.slow
def test_imagenet_resnet50_fp16_loss_scale_cpu(tmp_path, small_validation):
  """ResNet-50 one step on the CPU reference back end with a loss-scale wrapper."""
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "2", "-train_steps", "1", "-num_gpus", "0",
                   "-distribution_strategy", "off", "-dtype", "bf16", "-loss_scale", "128",
                   "-skip_eval"])
  assert stats["loss"] > 0


def test_fp16_rejected_for_cifar(tmp_path, small_validation):
  resnet_cifar_main.define_cifar_flags()
  with pytest.raises(ValueError):
    integration.run_synthetic(main=resnet_cifar_main.run, tmp_root=str(tmp_path),
                              extra_flags=_CIFAR_FLAGS + ["-distribution_strategy", "off",
                                                          "-dtype", "fp16"])


def test_mirrored_two_ranks_via_main_fanout(tmp_path, small_validation):
  """`--distribution_strategy mirrored --num_gpus 2` started directly fans out into 2 ranks."""
  import subprocess
  env = dict(os.environ, DTF_FORCE_CPU="1")
  r = subprocess.run([sys.executable, os.path.join(ROOT, "resnet_cifar_main.py"),
                      "--use_synthetic_data", "--distribution_strategy", "mirrored",
                      "--num_gpus", "2", "--batch_size", "8", "--train_steps", "2", "--skip_eval",
                      "--model_dir", str(tmp_path / "m"), "--log_steps", "1", "--num_packs", "2"],
                     env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-3000:]
  assert r.stderr.count("Run stats:") == 2          # both ranks finished
  assert "strategy=mirrored" in r.stderr


@pytest.mark.gpu
def test_imagenet_resnet50_native_end_to_end(tmp_path, small_validation):
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "16", "-train_steps", "3", "-num_gpus", "1", "-dtype", "bf16",
                   "-distribution_strategy", "one_device", "-log_steps", "1"], max_train=None)
  assert stats["loss"] > 0 and "avg_exp_per_second" in stats


@pytest.mark.gpu
def test_imagenet_real_data_native_end_to_end(tmp_path, small_validation):
  """ImageNet driver on a TFRecord tree (1024 + 128 shards of small synthetic JPEGs): worker
  processes decode / crop / resize, the device prefetcher stages batches, the native engine
  trains and evaluates."""
  import subprocess
  data = tmp_path / "imagenet"
  r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_fake_imagenet.py"),
                      str(data), "1", "72", "96"], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path), synth=False,
      extra_flags=["-data_dir", str(data), "-batch_size", "4", "-train_steps", "4", "-num_gpus", "1",
                   "-dtype", "bf16", "-distribution_strategy", "off", "-log_steps", "1"],
      max_train=None)
  assert stats["loss"] > 0 and "eval_loss" in stats and "avg_exp_per_second" in stats


@pytest.mark.gpu
def test_profile_steps_traces_native_kernels(tmp_path, small_validation):
  """`--profile_steps a,b` (`common.py:192-197`) on the native engine: the trace written to
  `model_dir` contains the hand-written kernels."""
  resnet_cifar_main.define_cifar_flags()
  integration.run_synthetic(
      main=resnet_cifar_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "16", "-train_steps", "12", "-num_gpus", "1", "-dtype", "bf16",
                   "-distribution_strategy", "off", "-skip_eval", "-profile_steps", "10,11",
                   "-model_dir", str(tmp_path / "prof")], max_train=None)
  trace = (tmp_path / "prof" / "profile_steps_10_11.trace.json").read_text()
  assert "conv_igemm_kernel" in trace and "bn_apply_kernel" in trace


def _free_ports(n):
  import socket
  socks = [socket.socket() for _ in range(n)]
  for s in socks:
    s.bind(("127.0.0.1", 0))
  ports = [s.getsockname()[1] for s in socks]
  for s in socks:
    s.close()
  return ports


def _run_pair(tmp_path, cmds, timeout=300):
  import subprocess
  env = dict(os.environ, OMP_NUM_THREADS="2")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    env.pop(k, None)
  procs = [subprocess.Popen([sys.executable] + c, env=env, cwd=str(tmp_path),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for c in cmds]
  outs = [p.communicate(timeout=timeout)[0].decode()[-3000:] for p in procs]
  assert all(p.returncode == 0 for p in procs), "\n=====\n".join(outs)
  return outs


def test_multi_worker_variant_drivers_on_localhost(tmp_path):
  """`resnet_cifar_main_dist.py` + `_dist_1.py`: the reference's two-terminal localhost
  rehearsal of multi_worker_mirrored (`README.md:120-124`)."""
  p0, p1 = _free_ports(2)
  common_args = ["--use_synthetic_data", "--num_gpus", "0", "--batch_size", "8", "--train_steps",
                 "2", "--skip_eval", "--worker_hosts", "localhost:%d,localhost:%d" % (p0, p1)]
  outs = _run_pair(tmp_path, [[os.path.join(ROOT, "resnet_cifar_main_dist.py")] + common_args,
                              [os.path.join(ROOT, "resnet_cifar_main_dist_1.py")] + common_args])
  assert all("Run stats" in o and "'loss'" in o for o in outs)


def test_parameter_server_variant_driver_on_localhost(tmp_path):
  """ps task + worker task of `resnet_cifar_main_dist_ps.py`; the ps task exits by itself."""
  p0, p1 = _free_ports(2)
  args = ["--use_synthetic_data", "--num_gpus", "0", "--batch_size", "8", "--train_steps", "3",
          "--skip_eval", "--ps_hosts", "localhost:%d" % p0, "--worker_hosts", "localhost:%d" % p1]
  script = os.path.join(ROOT, "resnet_cifar_main_dist_ps.py")
  outs = _run_pair(tmp_path, [[script] + args + ["--task_type", "ps"], [script] + args])
  assert "'loss'" in outs[1] and "Run stats" in outs[0]


def test_numbered_parameter_server_entry_points(tmp_path):
  """`resnet_cifar_main_dist_ps_remote_{0,1}.py` (role fixed by the file name, addresses from
  the environment) and the numbered ImageNet files `ps_server/..._ps_{0,1}.py` in a 1 ps +
  1 worker job with the trivial model."""
  import subprocess
  p0, p1 = _free_ports(2)
  args = ["--use_synthetic_data", "--num_gpus", "0", "--batch_size", "8", "--train_steps", "2",
          "--skip_eval"]
  os.environ["DTF_PS_HOST"], os.environ["DTF_WORKER_HOST"] = "localhost:%d" % p0, "localhost:%d" % p1
  try:
    outs = _run_pair(tmp_path, [
        [os.path.join(ROOT, "resnet_cifar_main_dist_ps_remote_0.py")] + args,
        [os.path.join(ROOT, "resnet_cifar_main_dist_ps_remote_1.py")] + args])
  finally:
    del os.environ["DTF_PS_HOST"], os.environ["DTF_WORKER_HOST"]
  assert "'loss'" in outs[1] and "Run stats" in outs[0]
  p0, p1 = _free_ports(2)
  os.environ["DTF_PS_HOSTS"], os.environ["DTF_WORKER_HOSTS"] = "localhost:%d" % p0, "localhost:%d" % p1
  try:
    outs = _run_pair(tmp_path, [
        [os.path.join(ROOT, "ps_server", "resnet_imagenet_main_dist_ps_%d.py" % k)] + args +
        ["--use_trivial_model"] for k in (0, 1)])
  finally:
    del os.environ["DTF_PS_HOSTS"], os.environ["DTF_WORKER_HOSTS"]
  assert "'loss'" in outs[1] and "Run stats" in outs[0]
  for k in range(16):
    assert os.path.exists(os.path.join(ROOT, "ps_server", "resnet_imagenet_main_dist_ps_%d.py" % k))


@pytest.mark.gpu
def test_cifar_real_data_native_end_to_end(tmp_path):
  """CIFAR driver on record files: C++ loader -> device prefetcher -> native executor, with
  evaluation and a Keras-layout checkpoint that loads back."""
  import numpy as np
  rng = np.random.default_rng(0)
  d = tmp_path / "cifar-10-batches-bin"
  d.mkdir()
  for name in ["data_batch_%d.bin" % i for i in range(1, 6)] + ["test_batch.bin"]:
    rec = rng.integers(0, 256, size=(64, 3073), dtype=np.uint8)
    rec[:, 0] = rng.integers(0, 10, size=64)
    (d / name).write_bytes(rec.tobytes())
  resnet_cifar_main.define_cifar_flags()
  old = dict(cifar_preprocessing.NUM_IMAGES)
  cifar_preprocessing.NUM_IMAGES.update(train=320, validation=64)
  try:
    flags_core.FLAGS(["prog", "--data_dir", str(d), "--model_dir", str(tmp_path / "m"),
                      "--batch_size", "32", "--train_epochs", "2", "--epochs_between_evals", "1",
                      "--num_gpus", "1", "--dtype", "bf16", "--distribution_strategy", "off",
                      "--log_steps", "5", "--export_dir", str(tmp_path / "export")])
    stats = resnet_cifar_main.run(flags_core.FLAGS)
  finally:
    cifar_preprocessing.NUM_IMAGES.update(old)
  assert stats["loss"] > 0 and 0.0 <= stats["accuracy_top_1"] <= 1.0 and stats["eval_loss"] > 0
  from distributed_tensorflow_amd.checkpoint import keras_h5
  w = keras_h5.read_weights(str(tmp_path / "export" / "resnet56.h5"))
  assert w["conv1/kernel"].shape == (3, 3, 3, 16) and np.isfinite(w["fc10/kernel"]).all()


@pytest.mark.gpu
def test_imagenet_fp16_flag_runs_native_engine_with_loss_scale(tmp_path, small_validation):
  """`--dtype fp16` (the reference's mixed-precision recipe, static loss scale 128) runs on
  the native engine: the scale is applied to the loss gradient and removed in the optimizer."""
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "16", "-train_steps", "4", "-num_gpus", "1", "-dtype", "fp16",
                   "-distribution_strategy", "off", "-log_steps", "1"], max_train=None)
  assert 0 < stats["loss"] < 20


@pytest.mark.gpu
def test_default_fp32_flag_runs_the_native_engine(tmp_path, small_validation):
  """The reference's default precision (`--dtype` unset = fp32, `resnet_imagenet_main.py:97-100`)
  trains and evaluates ResNet-50 on the native engine with exact fp32 matrix instructions -
  not on PyTorch ops."""
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path),
      extra_flags=["-batch_size", "16", "-train_steps", "3", "-num_gpus", "1",
                   "-distribution_strategy", "off", "-log_steps", "1"], max_train=None)
  assert 0 < stats["loss"] < 20 and "eval_loss" in stats
  from distributed_tensorflow_amd.train import model as model_mod
  assert model_mod.LAST_EXECUTOR_KIND == "hip"


@pytest.mark.gpu
def test_fp16_dynamic_loss_scale_and_trivial_model_on_the_native_engine(tmp_path, small_validation):
  """`--dtype fp16 --loss_scale dynamic` (device-side scaling) and `--use_trivial_model`."""
  (tmp_path / "a").mkdir()
  (tmp_path / "b").mkdir()
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path / "a"),
      extra_flags=["-batch_size", "16", "-train_steps", "4", "-num_gpus", "1", "-dtype", "fp16",
                   "-loss_scale", "dynamic", "-distribution_strategy", "off", "-skip_eval"],
      max_train=None)
  assert 0 < stats["loss"] < 20
  resnet_imagenet_main.define_imagenet_keras_flags()
  stats = integration.run_synthetic(
      main=resnet_imagenet_main.run, tmp_root=str(tmp_path / "b"),
      extra_flags=["-batch_size", "16", "-train_steps", "4", "-num_gpus", "1", "-dtype", "fp16",
                   "-use_trivial_model", "-distribution_strategy", "off", "-skip_eval"],
      max_train=None)
  assert "loss" in stats
  from distributed_tensorflow_amd.train import model as model_mod
  assert model_mod.LAST_EXECUTOR_KIND == "hip"
