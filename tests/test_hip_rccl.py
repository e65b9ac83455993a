"""The engine's own RCCL communicator (`ops/csrc/rccl_comm.cpp`, `parallel/rccl.py`) on one GPU.

A communicator of one rank exercises the whole call path - library loading, unique id,
`ncclCommInitRank`, every collective on a caller-chosen stream, collectives as launch-plan
entries and inside a captured hipGraph, the gradient reducer with `DTF_COMM=native` - with
results that are known: a sum over one rank is the identity.  Several ranks need several GPUs
(RCCL refuses two ranks on one device) and are not covered here.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
  from distributed_tensorflow_amd.parallel.rccl import NativeComm
  return NativeComm(0, 1, None, torch.device("cuda", 0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_collectives_of_one_rank_are_the_identity(comm, dtype):
  dev = torch.device("cuda", 0)
  x = torch.randn(1 << 20, device=dev).to(dtype)
  ref = x.clone()
  before = comm.collectives
  side = torch.cuda.Stream(dev)
  side.wait_stream(torch.cuda.current_stream(dev))
  for mode in ("oneshot", "direct", "auto"):
    comm.all_reduce(x, mode, stream=side.cuda_stream)
  comm.broadcast(x, 0, stream=side.cuda_stream)
  comm.reduce(x, 0, stream=side.cuda_stream)
  side.synchronize()
  assert torch.equal(x, ref)
  assert comm.collectives >= before + 5
  with pytest.raises(ValueError):
    comm.all_reduce(torch.zeros(4))                      # host tensor


def test_collective_as_plan_entry_and_inside_a_captured_graph(comm):
  """`allreduce` launch-plan entries: run eagerly, then captured together with kernels into one
  hipGraph and replayed - the collective becomes a node of the step's graph."""
  from distributed_tensorflow_amd.ops import native
  dev = torch.device("cuda", 0)
  g = torch.randn(850000, device=dev)                    # ResNet-56's 3.4 MB of gradients
  ref = g.clone()
  plan = native.Plan()
  plan.add("memset", dict(ptr=g.data_ptr(), nbytes=16), "zero_head", 0)
  plan.add("allreduce", comm.plan_entry(g, "oneshot"), "grads", 0)
  s = torch.cuda.Stream(dev)
  with torch.cuda.stream(s):
    h = native.stream_handle(dev)
    plan.run(0, plan.size(), h)
    s.synchronize()
    ref[:4] = 0
    assert torch.equal(g, ref)
    graph = plan.capture(0, plan.size(), h)
    g.copy_(torch.arange(g.numel(), device=dev, dtype=torch.float32))
    plan.replay(graph, h)
    s.synchronize()
  assert g[:4].abs().sum().item() == 0 and g[4].item() == 4.0 and g[-1].item() == g.numel() - 1


def test_reducer_on_the_native_communicator_in_a_one_rank_rehearsal(tmp_path):
  """bench.py's N > 1 step (segmented backward, bucket collectives on the communication stream,
  optimizer ordered behind them) with `DTF_COMM=native`: the collectives go through the
  extension's communicator, and the run ends with the same loss as through torch.distributed."""
  out = {}
  for kind in ("torch", "native"):
    env = dict(os.environ, DTF_COMM=kind, DTF_BENCH_REPORT_LOSS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "resnet20",
                        "--batch", "64", "--steps", "6", "--warmup", "3", "--rehearse_collectives",
                        "--num_packs", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out[kind] = json.loads(line)
  assert out["native"]["config"]["collectives"] == "native RCCL communicator"
  assert out["torch"]["config"]["collectives"] == "torch.distributed (nccl)"
  assert out["native"]["n_gpus"] == 1 and out["native"]["value"] > 0
  # the same training either way: the final losses of the two runs agree
  assert abs(out["native"]["config"]["final_loss"] - out["torch"]["config"]["final_loss"]) < 0.05


def test_single_bucket_gradient_is_reduced_inside_the_launch_plan():
  """Small-message policy with the native communicator: ResNet-20's gradient is one bucket, its
  all-reduce is the last entry of the backward plan (and a node of the captured graph)."""
  env = dict(os.environ, DTF_COMM="native")
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "resnet20",
                      "--batch", "64", "--steps", "12", "--warmup", "10", "--rehearse_collectives",
                      "--num_packs", "1"], env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-3000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
  assert out["config"]["collectives"] == "native RCCL communicator, all-reduce inside the launch plan"
  assert 0 < out["config"]["final_loss"] < 10
