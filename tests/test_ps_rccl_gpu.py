"""The parameter server's RCCL data plane on one GPU (one-rank communicators: a rank may send
to itself when the send and the matching receive share a group).

What a real job spreads over a PS rank and a worker runs here in one process, through the same
objects: bucket-wise bfloat16 pushes of a ResNet-50-sized gradient into a landing buffer
(`RcclPlane.exchange` on the push communicator's stream), the `[skip, factor]` trailer that the
PS side never reads on the host (`ParameterServerStrategy._finalize` with a device trailer: fused
SGD-momentum+L2 with the flag read by the kernel, moving statistics selected on the device,
outcome counted on the device), and the pull of the updated shard as bfloat16 on the other
communicator.  Checked against the same update computed with PyTorch operators."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _plane(dev):
  from distributed_tensorflow_amd.parallel.ps_transport import RcclPlane
  comm = types.SimpleNamespace(rank=0, world_size=1, device=dev, backend="nccl", rehearse=True,
                               store=lambda: None)
  return RcclPlane(comm, probe_against=torch.cuda.default_stream(dev))


def test_native_communicator_reports_health_and_runs_the_mesh_schedule():
  from distributed_tensorflow_amd.parallel.rccl import NativeComm
  dev = torch.device("cuda", 0)
  c = NativeComm(0, 1, None, dev)              # watchdog thread running
  c.check()                                    # healthy: no exception
  t = torch.arange(1024, dtype=torch.float32, device=dev)
  c.all_reduce(t, "mesh")                      # one rank: the schedule has no peers
  c.all_reduce(t, "oneshot")
  torch.cuda.synchronize()
  assert torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))
  c.check()
  from distributed_tensorflow_amd.parallel import rccl
  assert rccl._probe(c, 20.0)                  # the start-up all-reduce of a multi-rank job
  c.close()


@pytest.mark.timeout(120)
def test_push_apply_pull_through_the_rccl_plane_matches_the_reference_update():
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.engine.params import ParamStore
  from distributed_tensorflow_amd.parallel.parameter_server import ParameterServerStrategy, _ShardView
  from distributed_tensorflow_amd.parallel.ps_transport import PULL, PUSH
  dev = torch.device("cuda", 0)
  plane = _plane(dev)
  spec = models.resnet50()
  store = ParamStore(spec, dev, 0)
  store.slot("momentum")
  n, ns = store.num_elements, store.num_state_elements
  assert 25559081 <= n <= 25559081 + 64        # the flat buffer pads variables to 16 bytes
  buckets = store.make_buckets(bucket_bytes=16 << 20)
  assert len(buckets) >= 6
  g = torch.Generator(device="cpu").manual_seed(3)
  grads = (torch.randn(n, generator=g) * 1e-2).to(dev)
  new_state = torch.rand(ns, generator=g).to(dev)
  p0 = store.params.clone()
  lr, momentum, divisor = 0.1, 0.9, 4.0

  # ---- the strategy object's PS half, without a process group ---------------------------------
  ps = ParameterServerStrategy.__new__(ParameterServerStrategy)
  ps._host = {}
  ps.plane = plane
  ps.updates_applied = ps.updates_dropped = 0
  shard = _ShardView(store, 0, n)
  opt = types.SimpleNamespace(momentum=momentum)
  counters = torch.zeros(2, dtype=torch.float32, device=dev)
  main = torch.cuda.current_stream(dev)

  def one_push(skip_flag, factor):
    half = grads.to(torch.bfloat16)                       # worker: 16-bit copy of the gradients
    land = ps._buf(("ps_land", ("w", 1)), n, torch.bfloat16, dev)
    state_land = ps._buf(("ps_state", ("w", 1)), ns, torch.float32, dev)
    trailer_src = torch.tensor([skip_flag, factor], dtype=torch.float32, device=dev)
    trailer = ps._buf(("ps_trailer", 1), 2, torch.float32, dev)
    plane.order_after(PUSH, main)
    for bk in sorted(buckets, key=lambda b: b.index):     # pieces in the order backward completes them
      plane.exchange([(half[bk.start:bk.end], 0)], [(land[bk.start:bk.end], 0)], PUSH)
    works = plane.exchange([(new_state, 0), (trailer_src, 0)], [(state_land, 0), (trailer, 0)], PUSH)
    ev = torch.cuda.Event()
    ev.record(plane.stream(PUSH))
    main.wait_event(ev)                                   # PS: the update waits on the device
    p = dict(grads=land, state=state_land, trailer=trailer, lr=lr, divisor=divisor)
    ps._finalize(p, shard, opt, n, ns, True, False, dev, counters)
    return works

  # a dropped push changes nothing
  one_push(1.0, 1.0)
  torch.cuda.synchronize()
  assert torch.equal(store.params, p0)
  assert not torch.equal(store.state, new_state)
  # an applied push: w += v, v = m*v - lr * (g * factor / divisor + 2*lambda*w)
  one_push(0.0, 0.5)
  plane.check()
  gq = grads.to(torch.bfloat16).float() * (0.5 / divisor)
  v = -lr * (gq + store.l2_vector() * p0)
  want = p0 + v
  # pull: cast behind the update on the pull stream, transfer, widen
  plane.order_after(PULL, main)
  with torch.cuda.stream(plane.stream(PULL)):
    out = store.params.to(torch.bfloat16)
  got = torch.empty_like(out)
  for w in plane.exchange([(out, 0)], [(got, 0)], PULL):
    w.wait()
  torch.cuda.synchronize()
  assert torch.allclose(store.params, want, rtol=1e-5, atol=1e-7)
  assert torch.equal(store.state, new_state)
  assert torch.equal(got, want.to(torch.bfloat16)) or \
      (got.float() - want).abs().max() <= want.abs().max() * 2 ** -7
  assert counters.tolist() == [1.0, 1.0]                  # one applied, one dropped: counted on the GPU
  assert float(got.float().sum()) == pytest.approx(float(store.params.to(torch.bfloat16).float().sum()))
