"""Prints per-variable gradient agreement: HIP(bf16) vs torch(fp32) and torch(bf16) vs torch(fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.torch_executor import TorchExecutor
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor

def cos(a, b):
  a, b = a.flatten().double(), b.flatten().double()
  return (a @ b / (a.norm() * b.norm() + 1e-30)).item()

name = sys.argv[1] if len(sys.argv) > 1 else "resnet20"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
spec = getattr(models, name)()
dev = torch.device("cuda")
sts = [ParamStore(spec, dev, 0) for _ in range(3)]
hx = HipExecutor(spec, sts[0], batch, "bf16", use_graph=False)
t32 = TorchExecutor(spec, sts[1], batch, "fp32")
t16 = TorchExecutor(spec, sts[2], batch, "bf16")
g = torch.Generator().manual_seed(1)
H, W, C = spec.input_shape
x = (torch.randn(batch, H, W, C, generator=g) * 60 + 120).to(dev)
y = torch.randint(0, spec.num_classes, (batch,), generator=g).to(dev)
print("loss", hx.train_step(x, y)[0].item(), t32.train_step(x, y)[0].item(), t16.train_step(x, y)[0].item())
for e in sts[0].trainable:
  n = e.spec.name
  a, b, c = (s.view(n, "grad") for s in sts)
  print("%-28s hip/fp32 cos %.4f nr %.3f | torchbf16/fp32 cos %.4f nr %.3f | hip/torchbf16 cos %.4f" % (
      n, cos(a, b), (a.norm() / (b.norm() + 1e-30)).item(), cos(c, b), (c.norm() / (b.norm() + 1e-30)).item(), cos(a, c)))
