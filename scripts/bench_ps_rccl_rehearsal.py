"""Service costs of a parameter-server rank on the RCCL data plane, measured on ONE GPU.

A one-rank communicator can send to itself (send and matching receive in one group), so the PS
side of a ResNet-50 push / pull cycle runs here exactly as in a job - bucket-wise bfloat16 pieces
into a landing buffer on the push communicator's stream, the update (widen + fused
SGD-momentum+L2 with the trailer read on the device) ordered behind them, the pull's cast and
transfer on the other communicator's stream - only the bytes travel GPU -> same GPU instead of
over an xGMI link.  Prints the per-phase times and the timeline they imply for 1 ps + 7 workers
(`profiles/parameter_server_rehearsal.md`).

    python scripts/bench_ps_rccl_rehearsal.py [iterations]
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from distributed_tensorflow_amd import models  # noqa: E402
from distributed_tensorflow_amd.engine.params import ParamStore  # noqa: E402
from distributed_tensorflow_amd.parallel.parameter_server import ParameterServerStrategy, _ShardView  # noqa: E402
from distributed_tensorflow_amd.parallel.ps_transport import PULL, PUSH, RcclPlane  # noqa: E402

LINK_GBS = 153.0          # one xGMI link, one direction (datasheet); a worker owns one link to the ps


def main():
  iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
  dev = torch.device("cuda", 0)
  comm = types.SimpleNamespace(rank=0, world_size=1, device=dev, backend="nccl", rehearse=True,
                               store=lambda: None)
  plane = RcclPlane(comm, probe_against=torch.cuda.default_stream(dev))
  store = ParamStore(models.resnet50(), dev, 0)
  store.slot("momentum")
  n, ns = store.num_elements, store.num_state_elements
  buckets = sorted(store.make_buckets(bucket_bytes=16 << 20), key=lambda b: b.index)
  ps = ParameterServerStrategy.__new__(ParameterServerStrategy)
  ps._host, ps.plane, ps.updates_applied, ps.updates_dropped = {}, plane, 0, 0
  shard = _ShardView(store, 0, n)
  opt = types.SimpleNamespace(momentum=0.9)
  counters = torch.zeros(2, device=dev)
  main_s = torch.cuda.current_stream(dev)
  half = (torch.randn(n, device=dev) * 1e-3).to(torch.bfloat16)
  land = torch.empty_like(half)
  state_src, state_land = torch.rand(ns, device=dev), torch.empty(ns, device=dev)
  tr_src, tr = torch.tensor([0.0, 1.0], device=dev), torch.empty(2, device=dev)
  out16, got16 = torch.empty_like(half), torch.empty_like(half)

  def ev(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e

  t_push, t_apply, t_pull = [], [], []
  for it in range(iters + 3):
    plane.order_after(PUSH, main_s)
    a0 = ev(plane.stream(PUSH))
    for bk in buckets:
      plane.exchange([(half[bk.start:bk.end], 0)], [(land[bk.start:bk.end], 0)], PUSH)
    plane.exchange([(state_src, 0), (tr_src, 0)], [(state_land, 0), (tr, 0)], PUSH)
    a1 = ev(plane.stream(PUSH))
    main_s.wait_event(a1)
    b0 = ev(main_s)
    ps._finalize(dict(grads=land, state=state_land, trailer=tr, lr=0.1, divisor=1.0), shard, opt,
                 n, ns, True, False, dev, counters)
    b1 = ev(main_s)
    plane.order_after(PULL, main_s)
    c0 = ev(plane.stream(PULL))
    with torch.cuda.stream(plane.stream(PULL)):
      out16.copy_(store.params)
    plane.exchange([(out16, 0), (store.state, 0)], [(got16, 0), (state_land, 0)], PULL)
    c1 = ev(plane.stream(PULL))
    torch.cuda.synchronize()
    if it >= 3:
      t_push.append(a0.elapsed_time(a1))
      t_apply.append(b0.elapsed_time(b1))
      t_pull.append(c0.elapsed_time(c1))
  plane.check()
  med = lambda v: sorted(v)[len(v) // 2]
  mb16, mb_state = n * 2 / 1e6, ns * 4 / 1e6
  push, apply_, pull = med(t_push), med(t_apply), med(t_pull)
  link_ms = (mb16 + mb_state) / 1e3 / LINK_GBS * 1e3
  print("ResNet-50: %d parameters, %.1f MB as bfloat16 (+ %.2f MB of moving statistics), %d push pieces"
        % (n, mb16, mb_state, len(buckets)))
  print("measured on one GPU (self transfers), median of %d:" % iters)
  print("  push: %d pieces + statistics + trailer through the push communicator   %.3f ms (%.0f GB/s)"
        % (len(buckets), push, (mb16 + mb_state) / push))
  print("  update: widen to fp32 + fused SGD-momentum+L2, trailer read on the device  %.3f ms" % apply_)
  print("  pull: cast to bfloat16 + transfer + statistics                            %.3f ms" % pull)
  print("timeline of 1 ps + 7 workers (one xGMI link per worker, %.0f GB/s each way):" % LINK_GBS)
  print("  a push or a pull occupies the worker's link for %.2f ms; seven workers use seven links" % link_ms)
  print("  the ps GPU is busy %.2f ms per update and %.2f ms per pull (cast + copy engine): %.2f ms of "
        "every %.0f ms worker step for seven workers" % (apply_, pull, 7 * (apply_ + pull), 20.0))

if __name__ == "__main__":
  main()
