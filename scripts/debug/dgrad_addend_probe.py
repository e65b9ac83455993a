#!/usr/bin/env python
"""What the residual addend costs a data gradient: the pointwise layers whose data gradient adds the
shortcut's gradient (masked by the block input's ReLU bits), timed alone with the tuned tile -
plain, with the addend, with addend and mask - next to the bytes each form moves."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import autotune, native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom

SHAPES = [(56, 256, 64), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)]


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
  cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [None]
  for hw, cin, cout in SHAPES:
    g = ConvGeom(N, hw, hw, cin, 1, 1, cout, 1, (0, 0, 0, 0))
    dy = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16)
    wd = torch.randn(cin * g.cout_pad + 64, device=dev).to(torch.bfloat16) * 0.05
    dx = torch.empty(N, hw, hw, cin, dtype=torch.bfloat16, device=dev)
    add = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    mask = torch.randint(0, 255, (N * hw * hw * cin // 8,), dtype=torch.uint8, device=dev)
    mb = dx.numel() * 2 / 1e6
    for cfg in cfgs:
      row = []
      for kw in ({}, dict(addend=add.data_ptr()), dict(addend=add.data_ptr(), addend_mask=mask.data_ptr()),
                 dict(accumulate=True)):
        ds = [autotune.apply("conv", d) for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), **kw)]
        if cfg is not None:
          ds = [dict(d, cfg=cfg) for d in ds]
        row.append(timeit(lambda: [native.run_op("conv", d) for d in ds]))
      print("%dx%d dX %d ch (%.0f MB) <- dY %d ch, cfg %s: plain %.1f us, + addend %.1f, + masked addend %.1f, "
            "accumulate %.1f" % (hw, hw, cin, mb, cout, ds[0]["cfg"], *row), flush=True)


if __name__ == "__main__":
  main()
