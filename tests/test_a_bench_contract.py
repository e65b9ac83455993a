"""The benchmark entry point prints exactly one JSON line with the agreed fields.  (Named to
run before the tests that initialise the GPU in the pytest process: it starts GPU programs as
children.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--rehearse_collectives"]])
def test_bench_prints_one_json_line(extra):
  import torch
  if torch.cuda.is_initialized():
    pytest.skip("the GPU is already initialised in this process; child programs cannot be started")
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                      "--warmup", "1", "--batch", "32"] + extra,
                     env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
    assert key in out, key
  assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
  assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["dtype"] == "bf16"
  assert out["value"] > 0 and out["ms_per_step"] > 0
  assert out["config"]["executor"] == "hip" and out["config"]["model"].startswith("resnet50")
  # value is the whole-job rate: global batch * steps / time
  assert out["value"] == pytest.approx(
      out["config"]["global_batch"] * 1000.0 / out["ms_per_step"], rel=1e-2)


def test_bench_refuses_multi_gpu_without_launcher():
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                     env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode == 2 and "launcher" in r.stderr


def test_bench_eight_ranks_under_torchrun_on_cpu():
  """bench.py itself, launched exactly as the scaling run launches it (`torch.distributed.run`,
  eight ranks, 127.0.0.1), rehearsed on the CPU: gloo collectives, PyTorch-operator executor,
  a small model.  One JSON line from rank 0, `n_gpus` 8, whole-job rate over the global batch."""
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    env.pop(k, None)
  env["OMP_NUM_THREADS"] = "1"
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
         "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
         "--gpus", "8", "--steps", "2", "--warmup", "1", "--device", "cpu", "--model", "resnet20",
         "--batch", "2"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out["n_gpus"] == 8 and out["steps"] == 2 and out["warmup"] == 1
  assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8"
  assert out["dtype"] == "fp32" and out["config"]["executor"] == "torch"
  assert out["value"] == pytest.approx(16 * 1000.0 / out["ms_per_step"], rel=1e-2)
