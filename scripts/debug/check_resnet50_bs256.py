"""One full-size step (ResNet-50, batch 256) of the native engine against the PyTorch fp32
executor and against PyTorch's own bf16 execution: loss and per-variable gradient direction.
Deep in a randomly initialised network the bf16 gradient of a single step is noise dominated
in *any* bf16 execution; what matters is that the engine is as close to fp32 as the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.torch_executor import TorchExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
spec = models.resnet50()
stores = [ParamStore(spec, dev, 0) for _ in range(3)]
hx = HipExecutor(spec, stores[0], N, "bf16", stores[0].make_buckets(1), use_graph=False)
tx = TorchExecutor(spec, stores[1], N, "fp32")
bx = TorchExecutor(spec, stores[2], N, "bf16")
g = torch.Generator().manual_seed(1)
x = (torch.randn(N, 224, 224, 3, generator=g) * 60 + 120).to(dev)
y = torch.randint(0, 1001, (N,), generator=g).to(dev)
lh, _ = hx.train_step(x, y)
lt, _ = tx.train_step(x, y)
lb, _ = bx.train_step(x, y)
torch.cuda.synchronize()
print("loss: native %.4f  torch fp32 %.4f  torch bf16 %.4f" % (lh.item(), lt.item(), lb.item()))


def cos(a, b):
  a, b = a.flatten().double(), b.flatten().double()
  return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


rows = []
for e in stores[0].trainable:
  if e.spec.var != "kernel":
    continue
  n = e.spec.name
  gt = stores[1].view(n, "grad")
  rows.append((cos(stores[0].view(n, "grad"), gt), cos(stores[2].view(n, "grad"), gt), n))
print("%-28s %10s %10s" % ("kernel", "native", "torch bf16"))
for c, cl, n in rows[::6] + rows[-1:]:
  print("%-28s %10.4f %10.4f" % (n, c, cl))
import statistics
print("mean cosine vs fp32: native %.4f, torch bf16 %.4f; native >= library on %d of %d kernels"
      % (statistics.mean(r[0] for r in rows), statistics.mean(r[1] for r in rows),
         sum(r[0] >= r[1] - 0.02 for r in rows), len(rows)))
