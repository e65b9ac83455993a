#!/usr/bin/env python
"""Whole-rows kernels against the general kernels at the batch size of the training step, every
epilogue flavour, several launches each (a timing-dependent fault shows as a difference between
launches or against the other kernel family):  python scripts/debug/rows_stress.py [N=256]"""
import math
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
bad = 0


def close(a, b, what):
  global bad
  d = (a.float() - b.float()).abs()
  tol = 2.0 ** -6 * b.float().abs() + 2e-2
  n = int((d > tol).sum())
  if n:
    bad += 1
  print("   %-46s %s (max diff %.4g, %d elements off)" % (what, "ok" if not n else "DIFFERENT", d.max().item(), n), flush=True)


for (HW, C, cfgs, refcfg) in ((56, 64, (48, 50, 52, 53), 30), (28, 128, (49, 51), 30)):
  print("%dx%dx%d, N=%d" % (HW, HW, C, N))
  g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
  gen = torch.Generator().manual_seed(HW)
  x = T._bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  w = (torch.randn(3, 3, C, C, generator=gen) / math.sqrt(9 * C)).to(dev)
  wf, wd = T._pack(native, g, w, "fwd"), T._pack(native, g, w, "dgrad")
  y = T._bf(torch.randn(N, HW, HW, C, generator=gen) * 1.5 + 0.7).to(dev)
  add = T._bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  bits = torch.rand(N * HW * HW, C, generator=gen) > 0.4
  mask = (bits.view(-1, C // 8, 8).to(torch.int32) << torch.arange(8)).sum(-1).to(torch.uint8)
  mask = torch.cat([mask.flatten(), torch.zeros(4, dtype=torch.uint8)]).to(dev)
  del bits

  def fwd(cfg):
    out = torch.zeros(N, HW, HW, C, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(32, 2, C, device=dev)
    native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=st.data_ptr(),
                                stats_replicas=32))
    torch.cuda.synchronize()
    return out, st.sum(0)

  def dgrad(cfg, addend, stats):
    out = torch.zeros(N, HW, HW, C, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(32, 2, C, device=dev)
    for d in g.dgrad(x.data_ptr(), wd.data_ptr(), out.data_ptr(), cfg=cfg,
                     addend=add.data_ptr() if addend else None):
      if stats:
        d.update(stats=st.data_ptr(), stats_replicas=32, stats_y=y.data_ptr(), stats_mask=mask.data_ptr())
      native.run_op("conv", d)
    torch.cuda.synchronize()
    return out, st.sum(0)

  ref_f = fwd(refcfg)
  refs = {(a, s): dgrad(refcfg, a, s) for a in (False, True) for s in (False, True)}
  for cfg in cfgs:
    for rep in range(3):
      o, s = fwd(cfg)
      close(o, ref_f[0], "cfg %d forward + statistics, launch %d" % (cfg, rep))
      close(s / (N * HW * HW) * 100, ref_f[1] / (N * HW * HW) * 100, "   its statistics")
      for (a, st), (ro, rs) in refs.items():
        if a and not native.ext().conv_rows_serves(dict(g.dgrad(1, 2, 3, cfg=cfg, addend=4)[0])):
          continue
        o, s = dgrad(cfg, a, st)
        close(o, ro, "cfg %d data gradient%s%s, launch %d" % (cfg, " + addend" if a else "", " + statistics" if st else "", rep))
        if st:
          close(s / (N * HW * HW) * 100, rs / (N * HW * HW) * 100, "   its statistics")
print("FAULTS: %d" % bad)
