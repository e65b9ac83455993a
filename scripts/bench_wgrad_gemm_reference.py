#!/usr/bin/env python
"""Yardstick for the 1x1 weight gradients: dW = X^T dY as a plain library GEMM (torch.mm ->
hipBLASLt, bf16 operands, reduction over the N*H*W pixels) next to the native kernel with its
tuned tile / split count, ResNet-50 shapes at batch 256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import autotune, native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom

SHAPES = [(56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128),
          (28, 512, 256), (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)]


def timeit(fn, iters=20, warm=3):
  for _ in range(warm):
    fn()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  b.synchronize()
  return a.elapsed_time(b) / iters * 1000.0


def main():
  N = 256
  dev = torch.device("cuda")
  print("%-22s %10s %10s %8s" % ("HxW Cin->Cout", "library us", "native us", "ratio"))
  tl = tn = 0.0
  for hw, cin, cout in SHAPES:
    P = N * hw * hw
    x = torch.randn(P, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(P, cout, device=dev).to(torch.bfloat16)
    xt = x.t()
    t_lib = timeit(lambda: torch.mm(xt, dy))
    g = ConvGeom(N, hw, hw, cin, 1, 1, cout, 1, (0, 0, 0, 0))
    dw = torch.zeros(cin, cout, device=dev)
    d = autotune.apply("wgrad", g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr()))
    t_nat = timeit(lambda: native.run_op("wgrad", d))
    tl += t_lib
    tn += t_nat
    print("%-22s %10.1f %10.1f %8.2f" % ("%dx%d %d->%d" % (hw, hw, cin, cout), t_lib, t_nat, t_nat / t_lib), flush=True)
  print("%-22s %10.1f %10.1f %8.2f" % ("sum", tl, tn, tn / tl))


if __name__ == "__main__":
  main()
