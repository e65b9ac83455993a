"""Cost of issuing one all-reduce through torch.distributed in a one-rank RCCL group: host time
per call and the stall it puts on the compute stream (a chain of small kernels with and without
a collective between them).

    python scripts/bench_collective_overhead.py [numel]
"""
import os, sys, time, tempfile
import torch
import torch.distributed as dist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
path = os.path.join(tempfile.mkdtemp(), "store")
dist.init_process_group("nccl", store=dist.FileStore(path, 1), rank=0, world_size=1, device_id=dev)
g = torch.zeros(n, device=dev)
a = torch.zeros(1 << 16, device=dev)
comm_stream = torch.cuda.Stream(dev)


def chain(k=20):
  for _ in range(k):
    a.add_(1.0)


def v_none():
  chain(); chain()


def v_async():
  chain()
  w = dist.all_reduce(g, async_op=True)
  w.wait()
  chain()


def v_sync():
  chain()
  dist.all_reduce(g)
  chain()


def v_own_stream():
  chain()
  cur = torch.cuda.current_stream()
  comm_stream.wait_stream(cur)
  with torch.cuda.stream(comm_stream):
    dist.all_reduce(g)
  cur.wait_stream(comm_stream)
  chain()


for name, fn in (("no collective", v_none), ("async_op + wait", v_async), ("sync op", v_sync),
                 ("sync op on own stream", v_own_stream)):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  iters = 50
  t0 = time.perf_counter()
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  host = (time.perf_counter() - t0) / iters * 1e6
  torch.cuda.synchronize()
  print("%-24s host %7.1f us/iter   gpu %7.1f us/iter" % (name, host, e0.elapsed_time(e1) / iters * 1e3),
        flush=True)
dist.destroy_process_group()
