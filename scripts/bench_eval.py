"""Inference throughput of the native engine (forward pass with moving statistics + loss / accuracy).

    python scripts/bench_eval.py [resnet50|resnet56] [batch] [steps]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.data.synthetic import SyntheticDataset
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
spec = getattr(models, name)()
H, W, C = spec.input_shape
m = Model(spec, dtype="bf16", device=dev, executor="hip")
m.compile(SGD(0.1, momentum=0.9), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
ds = SyntheticDataset(H, W, C, spec.num_classes, N, dev, per_sample_random=True)
x, y = next(iter(ds))
for _ in range(5):
  m.test_on_batch(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
  out = m.test_on_batch(x, y)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s batch %d: %.3f ms per evaluation step, %.0f images/s" % (name, N, dt / steps * 1e3, N * steps / dt))
