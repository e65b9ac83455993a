"""Real multi-GPU runs over RCCL (skipped on machines with fewer than two GPUs; named to run
last).  The reference's 2-GPU cases are `resnet_imagenet_test.py:141-278`."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")


def _env():
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG", "MASTER_ADDR", "MASTER_PORT",
            "DTF_BACKEND", "DTF_REHEARSE_COLLECTIVES"):
    env.pop(k, None)
  return env


@pytest.mark.gpu
@needs_two_gpus
@pytest.mark.timeout(600)
def test_bench_on_two_gpus_over_rccl():
  """The scaling benchmark as it is started for N = 2: one JSON line, two replicas."""
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                      "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
                      os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                      "--batch", "64"], env=_env(), capture_output=True, text=True, timeout=540)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128
  assert out["config"]["parallelism"] == "dp2" and out["value"] > 0


@pytest.mark.gpu
@needs_two_gpus
@pytest.mark.timeout(600)
def test_two_replicas_hold_identical_variables_after_training(tmp_path):
  """Two native replicas, different data and different initial seeds: after the broadcast and
  three steps of bucketed all-reduce both hold the same variables."""
  script = tmp_path / "w.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\nimport torch\n" % ROOT + """
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.parallel import strategy as S, comm
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD
strat = S.MirroredStrategy(use_gpu=True, num_packs=3)
rank = strat.rank
m = Model(models.resnet20(), strategy=strat, dtype="bf16", device=strat.device, seed=rank)
assert m.executor_kind == "hip" and strat.comm.backend == "nccl"
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
g = torch.Generator().manual_seed(rank)
x = (torch.randn(16, 32, 32, 3, generator=g) * 50 + 100).to(strat.device)
y = torch.randint(0, 10, (16, 1), generator=g).float().to(strat.device)
losses = [m.train_on_batch(x, y)[0].item() for _ in range(10)]
torch.cuda.synchronize()
json.dump({"psum": m.store.params.double().sum().item(),
           "head": m.store.params[:64].tolist(), "losses": losses},
          open(os.path.join(%r, "r%%d.json" %% rank), "w"))
strat.barrier()
comm.shutdown()
""" % str(tmp_path))
  r = subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.launch", "--nproc", "2",
                      str(script)], env=dict(_env(), PYTHONPATH=ROOT), capture_output=True, text=True,
                     timeout=540)
  assert r.returncode == 0, r.stderr[-3000:]
  a, b = (json.load(open(tmp_path / ("r%d.json" % i))) for i in range(2))
  assert a["psum"] == b["psum"] and a["head"] == b["head"]
  assert all(l == l for l in a["losses"])           # finite
