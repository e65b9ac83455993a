"""Compares activations and activation gradients of the HIP engine with the torch executors."""
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
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
spec = getattr(models, name)()
dev = torch.device("cuda")
sts = [ParamStore(spec, dev, 0) for _ in range(3)]
hx = HipExecutor(spec, sts[0], batch, "bf16", use_graph=False)
t32 = TorchExecutor(spec, sts[1], batch, "fp32"); t32.debug_keep_activations = True
t16 = TorchExecutor(spec, sts[2], batch, "bf16"); t16.debug_keep_activations = True
g = torch.Generator().manual_seed(1)
H, W, C = spec.input_shape
x = (torch.randn(batch, H, W, C, generator=g) * 60 + 120).to(dev)
y = torch.randint(0, spec.num_classes, (batch,), generator=g).to(dev)
hx.train_step(x, y); t32.train_step(x, y); t16.train_step(x, y)
torch.cuda.synchronize()
n = 0
for op in spec.ops:
  o = op.output
  if o not in hx.acts or o not in t32.last_activations:
    continue
  a = hx.acts[o].float()
  r32 = t32.last_activations[o]; r16 = t16.last_activations[o]
  if r32.dim() == 4:
    r32v, r16v = r32.permute(0, 2, 3, 1), r16.permute(0, 2, 3, 1)
  else:
    r32v, r16v = r32, r16
  a = a[..., :r32v.shape[-1]]
  line = "%-26s act hip/32 %.5f lib16/32 %.5f" % (o, cos(a, r32v.float()), cos(r16v.float(), r32v.float()))
  if o in hx.act_grads and r32.grad is not None:
    ga = hx.act_grads[o].float()[..., :r32v.shape[-1]]
    g32 = r32.grad.permute(0, 2, 3, 1) if r32.dim() == 4 else r32.grad
    g16 = r16.grad.permute(0, 2, 3, 1) if r16.dim() == 4 else r16.grad
    line += " | grad hip/32 %.5f lib16/32 %.5f nr %.3f" % (cos(ga, g32.float()), cos(g16.float(), g32.float()), (ga.norm() / g32.norm()).item())
  print(line)
  n += 1
  if n >= limit: break
