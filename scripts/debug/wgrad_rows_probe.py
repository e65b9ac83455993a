#!/usr/bin/env python
"""Times the whole-rows weight-gradient kernel (cfg 192) on the 56x56x64 and 28x28x128 3x3 layers of ResNet-50:
with atomics, with plain stores (what the flush costs) and on a batch where every block has one tile
(what a tile costs end to end).  Run under rocprofv3 --pmc for the LDS counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom


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
  dev = torch.device("cuda")
  cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "192").split(",")]
  for N, HW, C in ((256, 56, 64), (64, 56, 64), (37, 56, 64), (256, 28, 128), (64, 28, 128)):
    g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
    x = torch.randn(N, HW, HW, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, HW, HW, C, device=dev).to(torch.bfloat16)
    dw = torch.zeros(3, 3, C, C, device=dev)
    for cfg in cfgs:
      d = g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg)
      t = timeit(lambda: native.run_op("wgrad", d))
      ta = timeit(lambda: native.run_op("wgrad", dict(d, atomic_flush=1)))
      tp = timeit(lambda: native.run_op("wgrad", dict(d, atomic_flush=1, debug_plain_store=1)))
      tn = float("nan")
      if cfg == 192:        # BatchNorm + ReLU of X inside the tile
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        tn = timeit(lambda: native.run_op("wgrad", dict(d, x_scale=sc.data_ptr(), x_shift=sh.data_ptr())))
      print("N=%d %dx%dx%d cfg %d: %.1f us (%.0f TFLOP/s); every block adding atomically %.1f us, plain "
            "stores instead %.1f us; normalising X in the tile %.1f us"
            % (N, HW, HW, C, cfg, t, 2 * N * HW * HW * C * 9 * C / t / 1e6, ta, tp, tn), flush=True)


def stem():
  """The 7x7 stride-2 stem (merged-tap form): whole-rows kernel against the tiles of the general one."""
  dev = torch.device("cuda")
  N = 256
  g = ConvGeom(N, 224, 224, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged")
  x = torch.randn(g.input_shape, device=dev).to(torch.bfloat16)
  dy = torch.randn(N, 112, 112, 64, device=dev).to(torch.bfloat16)
  dw = torch.zeros(7, 7, 3, 64, device=dev)
  for cfg, splits in ((192, 96), (1, 768), (1, 384), (3, 768)):
    d = g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=splits)
    t = timeit(lambda: native.run_op("wgrad", d))
    tp = timeit(lambda: native.run_op("wgrad", dict(d, debug_plain_store=1)))
    print("stem N=%d cfg %d splits %d: %.1f us, plain stores instead of atomics %.1f us" % (N, cfg, splits, t, tp), flush=True)


if __name__ == "__main__":
  if len(sys.argv) > 1 and sys.argv[1] == "stem":
    stem()
  else:
    main()
