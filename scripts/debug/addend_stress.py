#!/usr/bin/env python
"""Data gradients with an addend (the residual sum of the backward pass) and forward launches with
BatchNorm statistics at the batch size of the training step: every tile family against every other, several launches each - a timing-dependent
fault shows as a difference:  python scripts/debug/addend_stress.py [N=256]"""
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
faults = 0
# (H, Cin, K, Cout, stride, tiles): layers of ResNet-50 whose data gradient adds the shortcut's gradient
CASES = [(56, 64, 1, 256, 1, (0, 1, 5, 6, 10, 16, 17, 20, 21, 24, 25, 26, 30)),
         (28, 128, 1, 512, 1, (0, 4, 6, 16, 17, 24, 26, 28, 30)),
         (14, 256, 1, 1024, 1, (0, 6, 17, 18, 28, 29, 30, 31)),
         (56, 64, 3, 64, 1, (0, 1, 5, 21, 30)), (14, 256, 3, 256, 1, (0, 6, 30, 18))]
for (H, Cin, K, Cout, stride, cfgs) in CASES:
  pad = T.same_padding(H, K, stride) + T.same_padding(H, K, stride)
  g = ConvGeom(N, H, H, Cin, K, K, Cout, stride, pad)
  gen = torch.Generator().manual_seed(H + Cin)
  dy = T._bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) / math.sqrt(K * K * Cout)).to(dev)
  wd = T._pack(native, g, w, "dgrad")
  add = T._bf(torch.randn(N, H, H, Cin, generator=gen)).to(dev)
  bits = torch.rand(N * H * H, Cin, generator=gen) > 0.4
  mask = (bits.view(-1, Cin // 8, 8).to(torch.int32) << torch.arange(8)).sum(-1).to(torch.uint8)
  mask = torch.cat([mask.flatten(), torch.zeros(4, dtype=torch.uint8)]).to(dev)
  del bits
  for with_mask in (False, True):
    ref = None
    for cfg in cfgs:
      for rep in range(2):
        out = torch.zeros(N, H, H, Cin, dtype=torch.bfloat16, device=dev)
        try:
          for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), out.data_ptr(), cfg=cfg, addend=add.data_ptr(),
                           addend_mask=mask.data_ptr() if with_mask else None):
            native.run_op("conv", d)
        except ValueError:
          break
        torch.cuda.synchronize()
        if ref is None:
          ref = out
          continue
        dlt = (out.float() - ref.float()).abs()
        n = int((dlt > 2.0 ** -6 * ref.float().abs() + 2e-2).sum())
        if n:
          faults += 1
          print("%dx%dx%d k%d -> %d%s: tile %d launch %d differs from tile %d in %d elements (max %.3g)"
                % (H, H, Cin, K, Cout, " masked" if with_mask else "", cfg, rep, cfgs[0], n, dlt.max().item()), flush=True)
  # the forward launch of the same layer with the BatchNorm statistics in its epilogue
  x = T._bf(torch.randn(N, H, H, Cin, generator=gen)).to(dev)
  wf = T._pack(native, g, w, "fwd")
  ref = ref_st = None
  for cfg in cfgs:
    for rep in range(2):
      out = torch.zeros(g.output_shape, dtype=torch.bfloat16, device=dev)
      st = torch.zeros(32, 2, Cout, device=dev)
      try:
        native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=st.data_ptr(),
                                    stats_replicas=32))
      except ValueError:
        break
      torch.cuda.synchronize()
      st = st.sum(0)
      if ref is None:
        ref, ref_st = out, st
        continue
      dlt = (out.float() - ref.float()).abs()
      n = int((dlt > 2.0 ** -6 * ref.float().abs() + 2e-2).sum())
      ok_st = torch.allclose(st, ref_st, rtol=1e-3, atol=1e-3 * ref_st.abs().max().item() + 1e-2)
      if n or not ok_st:
        faults += 1
        print("%dx%dx%d k%d -> %d forward: tile %d launch %d differs from tile %d in %d elements, statistics %s"
              % (H, H, Cin, K, Cout, cfg, rep, cfgs[0], n, "ok" if ok_st else "OFF"), flush=True)
  print("%dx%dx%d k%d -> %d done" % (H, H, Cin, K, Cout), flush=True)
print("FAULTS: %d" % faults)
