"""Step time of the native ResNet-50 executor with different ways of issuing the per-bucket
collective (one-rank RCCL group): isolates what the reducer adds to a single-GPU step."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.train.optimizers import SGD

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
LATE = "late" in sys.argv     # form the process group after the executor created its streams


def init_pg():
  dist.init_process_group("nccl", store=dist.FileStore(os.path.join(tempfile.mkdtemp(), "s"), 1),
                          rank=0, world_size=1, device_id=dev)
  t = torch.zeros(8, device=dev)
  dist.all_reduce(t)            # creates the communicator and its streams
  torch.cuda.synchronize()


if not LATE:
  init_pg()
name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
packs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
graph = "graph" in sys.argv
spec = getattr(models, name)()
H, W, C = spec.input_shape
x = torch.randn(N, H, W, C, device=dev) * 50 + 100
y = torch.randint(0, spec.num_classes, (N,), device=dev)
store = ParamStore(spec, dev, 0)
buckets = store.make_buckets(packs)
ex = HipExecutor(spec, store, N, "bf16", buckets, use_graph=graph)
opt = SGD(0.01, momentum=0.9)
store.slot("momentum")
if LATE:
  for _ in range(2):
    ex.train_step(x, y, 1.0, None)
  torch.cuda.synchronize()
  init_pg()
comm_stream = torch.cuda.Stream(dev)
pending = []


def cb_none(b):
  pass


def cb_async(b):
  pending.append(dist.all_reduce(store.grads[b.start:b.end], async_op=True))


def cb_sync(b):
  dist.all_reduce(store.grads[b.start:b.end])


def cb_own(b):
  cur = torch.cuda.current_stream()
  comm_stream.wait_stream(cur)
  with torch.cuda.stream(comm_stream):
    dist.all_reduce(store.grads[b.start:b.end])
  pending.append(None)


def cb_events_only(b):
  cur = torch.cuda.current_stream()
  comm_stream.wait_stream(cur)
  pending.append(None)


def finish(kind):
  if kind == "async":
    for w in pending:
      w.wait()
  elif kind in ("own", "events_only"):
    torch.cuda.current_stream().wait_stream(comm_stream)
  pending.clear()


for kind, cb in (("no callback", None), ("none", cb_none), ("async", cb_async), ("sync", cb_sync),
                 ("own", cb_own), ("events_only", cb_events_only), ("no callback", None)):
  for it in range(12):
    if it == 4:
      torch.cuda.synchronize()
      t0 = time.perf_counter()
    ex.train_step(x, y, 1.0, cb)
    finish(kind)
    opt.apply_gradients(store, 1.0, ex)
  torch.cuda.synchronize()
  print("%-12s %.3f ms/step" % (kind, (time.perf_counter() - t0) / 8 * 1e3), flush=True)
dist.destroy_process_group()
from distributed_tensorflow_amd.ops import native
print("last lane probe ratios:", [round(v, 2) for v in native.ext().last_lane_probe()])
