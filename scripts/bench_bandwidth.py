"""Achievable streaming bandwidth on this GPU (reference points for the BatchNorm sweeps):
a device copy, a 2-read-1-write add, a read-only reduction and the framework's own bn kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import native

def t(fn, it=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(it): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / it * 1e-3

dev = torch.device("cuda")
for mb in (100, 400):
  n = mb * 1024 * 1024 // 2
  x = torch.randn(n, device=dev).to(torch.bfloat16); x2 = x.clone(); y = torch.empty_like(x)
  s = t(lambda: y.copy_(x));            print("%4d MB copy        %.2f TB/s" % (mb, 2 * n * 2 / s / 1e12))
  s = t(lambda: torch.add(x, x2, out=y)); print("%4d MB add 2r1w    %.2f TB/s" % (mb, 3 * n * 2 / s / 1e12))
  s = t(lambda: x.sum());               print("%4d MB sum (read)  %.2f TB/s" % (mb, n * 2 / s / 1e12))
  C = 256; P = n // C
  gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
  mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
  args = dict(y=x.data_ptr(), res=None, z=y.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
              mm=mm.data_ptr(), mv=mv.data_ptr(), P=P, C=C, eps=1e-5, relu=1)
  s = t(lambda: native.run_op("bn_infer", args)); print("%4d MB bn_apply    %.2f TB/s" % (mb, 2 * n * 2 / s / 1e12))
