"""Gradients of the two-lane plan (weight gradients on the side stream, ordered by fork / join
events) against the single-lane plan, same weights and batch, deterministic forward: any stale
read across the lanes would show up as a large difference in some gradient element; what is
left is the rounding noise of the fp32 atomics.

    python scripts/debug/check_lane_consistency.py [resnet50|resnet56] [batch] [steps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DTF_BN_FUSED_STATS"] = "0"
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
spec = getattr(models, name)()
H, W, C = spec.input_shape
g = torch.Generator().manual_seed(0)
batches = [((torch.randn(N, H, W, C, generator=g) * 60 + 120).to(dev),
            torch.randint(0, spec.num_classes, (N,), generator=g).to(dev)) for _ in range(steps)]
out = {}
for lanes in ("1", "0"):
  os.environ["DTF_WGRAD_STREAM"] = lanes
  st = ParamStore(spec, dev, 0)
  ex = HipExecutor(spec, st, N, "bf16", st.make_buckets(1), use_graph=False)
  gs = []
  for x, y in batches:
    ex.train_step(x, y)
    torch.cuda.synchronize()
    gs.append(st.grads.clone())
  out[lanes] = gs
  del ex
worst = 0.0
for a, b in zip(out["1"], out["0"]):
  d = (a - b).abs().max().item()
  scale = b.abs().max().item()
  rel = ((a - b).norm() / b.norm()).item()
  worst = max(worst, rel)
  print("max |diff| %.3e (max |grad| %.3e), relative L2 %.3e" % (d, scale, rel))
print("event flags %s: worst relative L2 difference %.3e -> %s"
      % (os.environ.get("DTF_EVENT_FLAGS", "default"), worst, "OK" if worst < 1e-3 else "MISMATCH"))
sys.exit(0 if worst < 1e-3 else 1)
