#!/usr/bin/env python
"""Rate of the BatchNorm apply sweep (read y, write z) by tensor shape: the shapes of ResNet-50 at
batch 256, each launched alone 20 times back to back (tensors of up to ~100 MB then live in the
256 MB memory-side cache: an upper bound for what they reach inside a step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import native


def t(fn, it=20):
  for _ in range(3):
    fn()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(it):
    fn()
  b.record()
  b.synchronize()
  return a.elapsed_time(b) / it * 1e-3


dev = torch.device("cuda")
N = 256
for hw, C in ((56, 64), (56, 256), (28, 128), (28, 512), (14, 256), (14, 1024), (7, 512), (7, 2048)):
  P = N * hw * hw
  x = torch.randn(P, C, device=dev).to(torch.bfloat16)
  y = torch.empty_like(x)
  gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
  mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
  args = dict(y=x.data_ptr(), res=None, z=y.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
              mm=mm.data_ptr(), mv=mv.data_ptr(), P=P, C=C, eps=1e-5, relu=1)
  s = t(lambda: native.run_op("bn_infer", args))
  mb = P * C * 2 / 1e6
  print("%2dx%2dx%4d  %6.1f MB  %6.1f us  %.2f TB/s" % (hw, hw, C, mb, s * 1e6, 2 * mb / s / 1e6), flush=True)
