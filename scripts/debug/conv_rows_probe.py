#!/usr/bin/env python
"""Times the whole-rows forward kernels (conv_rows.hip) on the 56x56x64 and 28x28x128 3x3 layers of
ResNet-50 with the BatchNorm statistics epilogue, per tile configuration:
    python scripts/debug/conv_rows_probe.py 52,53,48 [49,51]"""
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
  lists = [[int(c) for c in a.split(",")] for a in sys.argv[1:3]] or [[52]]
  shapes = [(256, 56, 64)] + ([(256, 28, 128)] if len(lists) > 1 else [])
  for (N, HW, C), cfgs in zip(shapes, lists):
    g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
    x = torch.randn(N, HW, HW, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(3, 3, C, C, device=dev) * 0.05)
    pd = dict(g.pack_fwd(), src_off=0, dst_off=0, first_block=0)
    wf = torch.zeros(pd["count"] + 8, dtype=torch.bfloat16, device=dev)
    desc = torch.frombuffer(bytearray(native.pack_desc_bytes([pd])), dtype=torch.uint8).to(dev)
    native.run_op("pack_weights", dict(params=w.contiguous().data_ptr(), packed=wf.data_ptr(), descs=desc.data_ptr(),
                                       ndesc=1, nblocks=native.pack_blocks(pd), dt=0))
    out = torch.zeros(N, HW, HW, C, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(32, 2, C, device=dev)
    ref = None
    for cfg in cfgs:
      d = g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=stats.data_ptr(), stats_replicas=32)
      t = timeit(lambda: native.run_op("conv", d))
      torch.cuda.synchronize()
      if ref is None:
        ref = out.clone()
      err = (out.float() - ref.float()).abs().max().item()
      print("N=%d %dx%dx%d cfg %d: %.1f us (%.0f TFLOP/s), max difference to the first cfg %.3g"
            % (N, HW, HW, C, cfg, t, 2 * N * HW * HW * C * 9 * C / t / 1e6, err), flush=True)


if __name__ == "__main__":
  main()
