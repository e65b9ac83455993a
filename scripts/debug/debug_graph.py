"""ResNet-20: eager launch vs hipGraph replay of the native step, side by side."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.train.optimizers import SGD
dev = torch.device("cuda")
spec = models.resnet20()
g = torch.Generator().manual_seed(2)
x = (torch.randn(32, 32, 32, 3, generator=g) * 60 + 120).to(dev)
y = torch.randint(0, 10, (32,), generator=g).to(dev)
for use_graph in (False, True):
  st = ParamStore(spec, dev, 0)
  ex = HipExecutor(spec, st, 32, "bf16", st.make_buckets(1), use_graph=use_graph)
  opt = SGD(0.05, momentum=0.9)
  losses = []
  for i in range(8):
    l, c = ex.train_step(x, y)
    gn = st.grads.norm().item()
    opt.apply_gradients(st, 1.0, ex)
    losses.append((round(l.item(), 4), round(gn, 4), int(c.item())))
  print("graph" if use_graph else "eager", losses)
