"""Per-variable gradient error of the native fp32 step against PyTorch fp32 (GPU) and fp64 (CPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.torch_executor import TorchExecutor

name, batch = sys.argv[1], int(sys.argv[2])
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
spec = getattr(models, name)()
dev = torch.device("cuda")
st_h, st_t = ParamStore(spec, dev, 7), ParamStore(spec, dev, 7)
hx = HipExecutor(spec, st_h, batch, "fp32", st_h.make_buckets(3), use_graph=False)
tx = TorchExecutor(spec, st_t, batch, "fp32", st_t.make_buckets(3))
g = torch.Generator(device="cpu").manual_seed(3)
H, W, C = spec.input_shape
x = (torch.randn(batch, H, W, C, generator=g) * 60 + 120).cuda()
y = torch.randint(0, spec.num_classes, (batch,), generator=g).cuda()
lh, _ = hx.train_step(x, y, 1.0)
lt, _ = tx.train_step(x, y, 1.0)
torch.cuda.synchronize()
print("loss", lh.item(), lt.item())
ref64 = None
if len(sys.argv) > 3:
  # third opinion: the same step with PyTorch's CPU kernels (different implementation, fp32)
  st_d = ParamStore(spec, torch.device("cpu"), 7)
  dx = TorchExecutor(spec, st_d, batch, "fp32", st_d.make_buckets(3))
  ld, _ = dx.train_step(x.cpu(), y.cpu(), 1.0)
  print("loss cpu", ld.item())
  ref64 = st_d
rows = []
for e in st_h.trainable:
  gh, gt = st_h.view(e.spec.name, "grad"), st_t.view(e.spec.name, "grad")
  r = ((gh - gt).norm() / gt.norm().clamp_min(1e-12)).item()
  row = [e.spec.name, r]
  if ref64 is not None:
    gd = ref64.view(e.spec.name, "grad").float().cuda()
    row += [((gh - gd).norm() / gd.norm()).item(), ((gt - gd).norm() / gd.norm()).item()]
  rows.append(row)
rows.sort(key=lambda r: -r[1])
print('columns: native-vs-gpu_torch  native-vs-cpu_torch  gpu_torch-vs-cpu_torch')
import collections
by=collections.defaultdict(list)
for r in rows: by[r[0].split('/')[-1]].append(r[1:])
for k,v in by.items(): print(k, 'max', [max(c) for c in zip(*v)], 'median', [sorted(c)[len(c)//2] for c in zip(*v)])
for r in rows[:8]:
  print(" ".join("%-34s" % r[0] if i == 0 else "%.3e" % v for i, v in enumerate(r)))
# activation comparison for the first ops
tx.debug_keep_activations = True
tx.train_step(x, y, 1.0)
for op in spec.ops[:6]:
  a = hx.acts[op.output].float()
  b = tx.last_activations[op.output].permute(0, 2, 3, 1).float() if tx.last_activations[op.output].dim() == 4 else tx.last_activations[op.output].float()
  a = a[..., :b.shape[-1]] if a.shape != b.shape and a.dim() == b.dim() else a
  if a.shape == b.shape:
    print(op.name, op.kind, "act rel", ((a - b).norm() / b.norm()).item())
  else:
    print(op.name, op.kind, "shape", tuple(a.shape), tuple(b.shape))
