"""ResNet-50: eager launch vs hipGraph replay of the native step, with timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = getattr(models, name)()
H, W, C = spec.input_shape
g = torch.Generator().manual_seed(2)
x = (torch.randn(B, H, W, C, generator=g) * 60 + 120).to(dev)
y = torch.randint(0, spec.num_classes, (B,), generator=g).to(dev)
st = ParamStore(spec, dev, 0)
ex = HipExecutor(spec, st, B, "bf16", st.make_buckets(1), use_graph=True)
outs = []
for i in range(4):
  l, c = ex.train_step(x, y)
  torch.cuda.synchronize()
  outs.append((l.item(), st.grads.clone(), {k: v.clone() for k, v in ex.act_grads.items()}, {k: v.clone() for k, v in ex.acts.items()}))
  print("call", i, "loss", l.item(), "gradnorm", st.grads.norm().item(), "finite", torch.isfinite(st.grads).all().item())
ref = outs[0]
for i in (1, 2, 3):
  o = outs[i]
  worst = None
  for e in st.trainable:
    a = ref[1][e.offset:e.offset + e.size]; b = o[1][e.offset:e.offset + e.size]
    d = (a - b).abs().max().item() / (a.abs().max().item() + 1e-20)
    if worst is None or d > worst[1]:
      worst = (e.spec.name, d)
  print("call", i, "worst grad rel diff", worst)
  for op in spec.ops:
    k = op.output
    a, b = ref[3][k].float(), o[3][k].float()
    d = (a - b).abs().max().item()
    if d > 0:
      print("  first act diff at", k, d); break
  for op in reversed(spec.ops):
    k = op.output
    if k in ref[2]:
      a, b = ref[2][k].float(), o[2][k].float()
      d = (a - b).abs().max().item() / (a.abs().max().item() + 1e-20)
      if d > 1e-2 or not torch.isfinite(b).all():
        print("  first act-grad diff (backward order) at", k, d, torch.isfinite(b).all().item()); break
