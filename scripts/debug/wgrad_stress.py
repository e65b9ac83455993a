#!/usr/bin/env python
"""Whole-rows weight gradient against the general kernel at batch 256, with and without BatchNorm
inside the tile, several launches each; the deterministic mode (one split) must repeat bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom
import test_hip_kernels as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
faults = 0
for HW, C in ((56, 64), (28, 128)):
  g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
  gen = torch.Generator().manual_seed(C)
  y = T._bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  dy = T._bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  scale = (torch.rand(C, generator=gen) + 0.5).to(dev)
  shift = (torch.randn(C, generator=gen) * 0.5 + 0.3).to(dev)
  a = T._affine_relu(y, scale, shift)

  def run(x, cfg, splits, norm):
    dw = torch.zeros(3, 3, C, C, device=dev)
    d = g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=splits)
    if norm:
      d.update(x_scale=scale.data_ptr(), x_shift=shift.data_ptr())
    native.run_op("wgrad", d)
    torch.cuda.synchronize()
    return dw

  ref = run(a, 16, 22, False)
  first = None
  for rep in range(3):
    for what, dw in (("plain", run(a, 192, 22, False)), ("normalising X", run(y, 192, 22, True)),
                     ("one split", run(a, 192, 1, False)), ("one split, normalising X", run(y, 192, 1, True))):
      err = (dw - ref).abs().max().item() / ref.abs().max().item()
      ok = err < 2e-4
      if "one split" in what:
        if first is None:
          first = dw
        ok = ok and torch.equal(dw, first)
      faults += not ok
      print("%dx%dx%d launch %d %-26s max difference / max |dW| = %.2e %s" % (HW, HW, C, rep, what, err, "ok" if ok else "FAULT"),
            flush=True)
  del first
print("FAULTS: %d" % faults)
