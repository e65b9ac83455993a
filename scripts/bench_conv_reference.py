"""MIOpen (through torch, bf16 channels_last) on every distinct convolution shape of ResNet-50
bs256: forward, data gradient and weight gradient, next to the native kernels' times from
scripts/bench_layers.py.  A yardstick, not a dependency.

    python scripts/bench_conv_reference.py [batch]
"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distributed_tensorflow_amd import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True
spec = models.resnet50()
shapes = collections.OrderedDict()
for op in spec.ops:
  if op.kind != "conv":
    continue
  a = op.attrs
  h, w, c = spec.shapes[op.inputs[0]]
  shapes.setdefault((h, w, c, a["kh"], a["cout"], a["stride"], tuple(a["pad"])), []).append(op.name)


def timeit(fn, iters=10):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3


tot = collections.Counter()
print("%-30s %3s | %9s %9s %9s" % ("shape", "n", "fwd us", "dgrad us", "wgrad us"))
for (h, w, c, k, cout, stride, pad), names in shapes.items():
  x = torch.randn(N, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  wt = (torch.randn(cout, c, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  pt, pb, pl, pr = pad
  if pt != pb or pl != pr:              # TF SAME with stride 2: pad bottom / right only
    x = F.pad(x, (pl, pr, pt, pb)).contiguous(memory_format=torch.channels_last)
    padding = (0, 0)
  else:
    padding = (pt, pl)
  y = F.conv2d(x, wt, None, stride, padding)
  dy = torch.randn_like(y)
  t_f = timeit(lambda: F.conv2d(x, wt, None, stride, padding))
  def bwd(mask):
    return torch.ops.aten.convolution_backward(dy, x, wt, None, (stride, stride), padding, (1, 1),
                                               False, (0, 0), 1, mask)
  t_d = timeit(lambda: bwd((True, False, False))) if c > 4 else 0.0
  t_w = timeit(lambda: bwd((False, True, False)))
  n = len(names)
  tot["f"] += n * t_f; tot["d"] += n * t_d; tot["w"] += n * t_w
  print("%-30s %3d | %9.1f %9.1f %9.1f" % ("%dx%dx%d k%d s%d ->%d" % (h, w, c, k, stride, cout), n,
                                           t_f, t_d, t_w), flush=True)
print("totals per step (ms): fwd %.2f dgrad %.2f wgrad %.2f" % (tot["f"] / 1e3, tot["d"] / 1e3, tot["w"] / 1e3))
