#!/usr/bin/env python
"""Pointwise layers alone, with and without BatchNorm + ReLU on their operand fragments: the forward
launch (`in_scale`) and the weight gradient (`x_scale`) of the bn2b -> conv2c shapes of ResNet-50.

    python scripts/debug/frag_probe.py [batch = 256]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from distributed_tensorflow_amd.ops import native  # noqa: E402
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom, round_up  # noqa: E402


def timed(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1000.0 / n


def main():
  N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
  dev = torch.device("cuda")
  ext = native.ext()
  print("shape                fwd cfg: plain -> fragments (us)   wgrad cfg/splits: plain -> fragments (us)")
  for hw, cin, cout in ((56, 64, 256), (28, 128, 512), (14, 256, 1024), (7, 512, 2048)):
    g = ConvGeom(N, hw, hw, cin, 1, 1, cout, 1, (0, 0, 0, 0))
    y = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
    z = torch.relu(y)
    scale = torch.rand(cin, device=dev) + 0.5
    shift = torch.randn(cin, device=dev) * 0.5
    w = (torch.randn(cout, round_up(cin, 8), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(g.output_shape, dtype=torch.bfloat16, device=dev)
    dy = torch.randn(g.output_shape, device=dev).to(torch.bfloat16)
    dw = torch.zeros(1, 1, cin, cout, device=dev)
    stats = torch.zeros(32, 2, cout, device=dev)
    line = "%dx%dx%d -> %d" % (hw, hw, cin, cout)
    for cfg in (17, 30, 31):
      f0 = g.fwd(z.data_ptr(), w.data_ptr(), out.data_ptr(), cfg=cfg, stats=stats.data_ptr(), stats_replicas=32)
      f1 = dict(g.fwd(y.data_ptr(), w.data_ptr(), out.data_ptr(), cfg=cfg, stats=stats.data_ptr(), stats_replicas=32),
                in_scale=scale.data_ptr(), in_shift=shift.data_ptr())
      line += "   c%d: %.1f -> %.1f" % (cfg, timed(lambda: native.run_op("conv", f0)),
                                        timed(lambda: native.run_op("conv", f1)))
    for cfg, splits in ((1, 96), (16, 96), (32 + 0, 48), (64 + 2, 96)):
      w0 = g.wgrad(z.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=splits)
      w1 = dict(g.wgrad(y.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=splits),
                x_scale=scale.data_ptr(), x_shift=shift.data_ptr())
      line += "   w%d/%d: %.1f -> %.1f" % (cfg, splits, timed(lambda: native.run_op("wgrad", w0)),
                                            timed(lambda: native.run_op("wgrad", w1)))
    print(line, flush=True)
  del ext


if __name__ == "__main__":
  main()
