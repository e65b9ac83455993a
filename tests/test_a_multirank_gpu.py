"""Multi-rank run of the native executor on one GPU (runs first: it starts child processes,
which has to happen before this pytest process initialises the GPU)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
@pytest.mark.timeout(700)     # the first GPU start-up on a fresh machine can take minutes
def test_two_ranks_native_executor_with_overlapped_allreduce(tmp_path):
  """Two ranks in one gloo group, both training with the native executor on the same GPU:
  exercises the segmented backward + bucket-ready path of the native executor under a real
  gradient reducer with a real peer that runs the same kernels (the driver time-slices the
  GPU between the two processes; profiles/two_process_stall.md)."""
  import subprocess, textwrap, json
  if torch.cuda.is_initialized():
    pytest.skip("the GPU is already initialised in this process; child programs cannot be started")
  script = tmp_path / "w.py"
  script.write_text("import sys, os, json, faulthandler\nsys.path.insert(0, %r)\nimport torch\n"
                    "faulthandler.dump_traceback_later(120, exit=True)\n" % ROOT + textwrap.dedent("""
    from distributed_tensorflow_amd import models
    from distributed_tensorflow_amd.parallel import strategy as S
    from distributed_tensorflow_amd.train.model import Model
    from distributed_tensorflow_amd.train.optimizers import SGD
    rank = int(os.environ["RANK"])
    strat = S.MirroredStrategy(use_gpu=True, num_packs=3)
    assert strat.rank == rank and strat.device.type == "cuda"
    m = Model(models.resnet20(), strategy=strat, dtype="bf16", device=strat.device, seed=rank)
    assert m.executor_kind == "hip"
    m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
    g = torch.Generator().manual_seed(rank)
    x = (torch.randn(16, 32, 32, 3, generator=g) * 50 + 100).to(strat.device)
    y = torch.randint(0, 10, (16, 1), generator=g).float().to(strat.device)
    losses = [m.train_on_batch(x, y)[0].item() for _ in range(4)]
    torch.cuda.synchronize()
    json.dump({"psum": m.store.params.double().sum().item(), "losses": losses,
               "ncoll": m._grad_reducer().collectives_launched},
              open(os.path.join(os.environ["DTF_TEST_OUT"], "r%d.json" % rank), "w"))
    strat.barrier()
  """))
  sys.path.insert(0, ROOT)
  from distributed_tensorflow_amd import launch
  code = launch.launch([str(script)], 2, log_dir=str(tmp_path / "logs"),
                       env_extra={"DTF_TEST_OUT": str(tmp_path), "DTF_BACKEND": "gloo",
                                  "LOCAL_RANK": "0"}, timeout_s=300)
  logs = "".join((tmp_path / "logs" / ("log%d.log" % i)).read_text()[-2000:] for i in range(2))
  assert code == 0, logs
  r = [json.load(open(tmp_path / ("r%d.json" % i))) for i in range(2)]
  # both replicas applied the same reduced gradients to the same broadcast variables
  assert r[0]["psum"] == pytest.approx(r[1]["psum"], rel=1e-6)
  assert r[0]["ncoll"] == 4 * 3
  assert r[0]["losses"][-1] < r[0]["losses"][0]


@pytest.mark.gpu
@pytest.mark.timeout(400)
@pytest.mark.parametrize("alg", [None, "ring", "hierarchical_copy"])
def test_one_rank_rccl_rehearsal_of_the_overlapped_reducer(tmp_path, alg):
  """The multi-GPU step (segmented backward, per-bucket RCCL collective on the communication
  stream, optimizer ordered after it) in a one-rank RCCL group on the GPU: real `nccl` backend
  calls, real stream hand-offs, and the same trajectory as the plain single-GPU step."""
  import subprocess, json
  if torch.cuda.is_initialized():
    pytest.skip("the GPU is already initialised in this process; child programs cannot be started")
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from test_distributed import _rehearsal_script
  script = _rehearsal_script(tmp_path, "cuda", "bf16", alg)
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG", "DTF_BACKEND"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                     timeout=360)
  assert r.returncode == 0, r.stderr[-3000:]
  out = json.loads((tmp_path / "out.json").read_text())
  assert out["rehearsed"]["backend"] == "nccl"
  per_bucket = 1 if alg is None else 2
  # 10 steps: eager launches, the capture step and hipGraph replays of the backward segments
  assert out["rehearsed"]["ncoll"] == 10 * 3 * per_bucket
  # fp32 atomics make the two runs differ in the last bits only
  assert out["rehearsed"]["losses"] == pytest.approx(out["plain"]["losses"], rel=5e-2)
  assert out["rehearsed"]["losses"][-1] < out["rehearsed"]["losses"][0]
