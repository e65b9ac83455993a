"""Library GEMM (torch.mm -> hipBLASLt / rocBLAS) on the 1x1-convolution shapes of ResNet-50 bs256,
as a yardstick for the pointwise mode of conv_igemm (which additionally fuses BatchNorm statistics /
masked residual gradients into its epilogue).

    python scripts/bench_gemm_reference.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom
from distributed_tensorflow_amd.ops import autotune

dev = torch.device("cuda")
N = 256
shapes = [(56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128),
          (28, 512, 256), (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)]


def timeit(fn, iters=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e3


print("%-22s %10s %10s %8s" % ("HxW Cin->Cout", "library us", "native us", "ratio"))
tl = tn = 0.0
for hw, cin, cout in shapes:
  M = N * hw * hw
  x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(cout, cin, device=dev) * 0.05).to(torch.bfloat16)
  out = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
  t_lib = timeit(lambda: torch.mm(x, w.t(), out=out))
  g = ConvGeom(N, hw, hw, cin, 1, 1, cout, 1, (0, 0, 0, 0))
  o2 = torch.empty(g.output_shape, device=dev, dtype=torch.bfloat16)
  d = autotune.apply("conv", g.fwd(x.data_ptr(), w.data_ptr(), o2.data_ptr()))
  t_nat = timeit(lambda: native.run_op("conv", d))
  tl += t_lib; tn += t_nat
  print("%-22s %10.1f %10.1f %8.2f" % ("%dx%d %d->%d" % (hw, hw, cin, cout), t_lib, t_nat, t_nat / t_lib))
print("%-22s %10.1f %10.1f %8.2f" % ("sum", tl, tn, tn / tl))
