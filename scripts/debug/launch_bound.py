"""Is a training step bound by the host's launch rate or by the GPU?  Times how long the host
needs to enqueue K steps (no synchronisation) against the time until the GPU has finished them.

    python scripts/debug/launch_bound.py [resnet56|resnet50] [batch] [steps]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.train.optimizers import SGD

name = sys.argv[1] if len(sys.argv) > 1 else "resnet56"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
spec = getattr(models, name)()
H, W, C = spec.input_shape
st = ParamStore(spec, dev, 0)
ex = HipExecutor(spec, st, N, "bf16", st.make_buckets(1), use_graph=False)
opt = SGD(0.01, momentum=0.9)
x = torch.randn(N, H, W, C, device=dev) * 60 + 120
y = torch.randint(0, spec.num_classes, (N,), device=dev)
for _ in range(10):
  ex.train_step(x, y)
  opt.apply_gradients(st, 1.0, ex)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
  ex.train_step(x, y)
  opt.apply_gradients(st, 1.0, ex)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
one = []
for _ in range(20):                      # empty queue: no back-pressure on the host
  torch.cuda.synchronize()
  a = time.perf_counter()
  ex.train_step(x, y)
  opt.apply_gradients(st, 1.0, ex)
  one.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("host time to enqueue one step into an empty queue: median %.3f ms" % (sorted(one)[10] * 1e3))
print("%s bs%d: host enqueue %.3f ms/step, GPU complete %.3f ms/step, %d fwd + %d bwd launches"
      % (name, N, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3, ex.fwd_plan.size(),
         ex.bwd_plan.size()))
