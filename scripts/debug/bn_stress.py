#!/usr/bin/env python
"""BatchNorm forward / backward sweeps on tensors of the training step's size against the fp32
formulas, element by element (a timing-dependent fault shows as isolated elements off), two launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from distributed_tensorflow_amd.ops import native
import test_hip_kernels as T

dev = torch.device("cuda")
faults = 0
for (P, C, residual) in ((256 * 56 * 56, 64, False), (256 * 56 * 56, 256, True), (256 * 28 * 28, 512, True),
                         (256 * 14 * 14, 1024, True), (256 * 7 * 7, 2048, True), (256 * 28 * 28, 128, False)):
  gen = torch.Generator().manual_seed(C)
  y = T._bf(torch.randn(P, C, generator=gen) * 2 + 0.5).to(dev)
  res = T._bf(torch.randn(P, C, generator=gen)).to(dev) if residual else None
  dz = T._bf(torch.randn(P, C, generator=gen)).to(dev)
  gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
  beta = (torch.randn(C, generator=gen) * 0.1).to(dev)
  yf = y.float()
  mean, var = yf.mean(0), yf.var(0, unbiased=False)
  invstd = torch.rsqrt(var + 1e-5)
  zr = (yf - mean) * invstd * gamma + beta
  if residual:
    zr += res.float()
  zr = torch.relu(zr)
  for rep in range(2):
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    saved = torch.zeros(4 * C, device=dev)
    ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
    z = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    mask = torch.zeros((P * C // 8 + 3) // 4 * 4, dtype=torch.uint8, device=dev)
    native.run_op("bn_fwd_train", dict(
        y=y.data_ptr(), res=res.data_ptr() if residual else None, z=z.data_ptr(), mask=mask.data_ptr(),
        gamma=gamma.data_ptr(), beta=beta.data_ptr(), mm=mm.data_ptr(), mv=mv.data_ptr(),
        save_mean=saved.data_ptr(), save_invstd=saved.data_ptr() + 4 * C, scale=saved.data_ptr() + 8 * C,
        shift=saved.data_ptr() + 12 * C, ws=ws.data_ptr(), P=P, C=C, eps=1e-5, momentum=0.9, relu=1))
    torch.cuda.synchronize()
    zq = z.float()
    off_f = int(((zq - zr).abs() > 2.0 ** -6 * zr.abs() + 2e-2).sum())
    bits = (zq > 0).view(P, C // 8, 8).to(torch.int32)
    expect = (bits << torch.arange(8, device=dev)).sum(-1).to(torch.uint8)
    off_m = int((mask[:P * C // 8].view(P, C // 8) != expect).sum())
    del bits, expect
    g = dz.float() * (zq > 0)
    xhat = (yf - mean) * invstd
    dgamma_ref, dbeta_ref = (g * xhat).sum(0), g.sum(0)
    dy_ref = gamma * invstd * (g - dbeta_ref / P - xhat * dgamma_ref / P)
    del xhat
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    dres = torch.empty(P, C, dtype=torch.bfloat16, device=dev) if residual else None
    native.run_op("bn_bwd", dict(dz=dz.data_ptr(), y=y.data_ptr(), z=None, mask=mask.data_ptr(),
                                 mean=saved.data_ptr(), invstd=saved.data_ptr() + 4 * C, gamma=gamma.data_ptr(),
                                 dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), dy=dy.data_ptr(),
                                 dres=dres.data_ptr() if residual else None, ws=ws.data_ptr(), P=P, C=C))
    torch.cuda.synchronize()
    off_b = int(((dy.float() - dy_ref).abs() > 2.0 ** -6 * dy_ref.abs() + 2e-2).sum())
    off_r = int((dres.float() != T._bf(g).float()).sum()) if residual else 0
    sums_ok = torch.allclose(dgamma, dgamma_ref, rtol=1e-2, atol=0.5) and torch.allclose(dbeta, dbeta_ref, rtol=1e-2, atol=0.5)
    bad = off_f + off_m + off_b + off_r + (not sums_ok)
    faults += bad > 0
    print("P=%d C=%d%s launch %d: forward %d, mask bytes %d, backward %d, residual gradient %d elements off, sums %s"
          % (P, C, " +res" if residual else "", rep, off_f, off_m, off_b, off_r, "ok" if sums_ok else "OFF"), flush=True)
    del g, dy_ref, zq
print("FAULTS: %d" % faults)
