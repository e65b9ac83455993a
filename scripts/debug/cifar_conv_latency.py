"""Latency of the three CIFAR ResNet convolution shapes (bs128) per tile configuration, launched
back to back on one stream from a native plan (what the dependent chain of a step sees)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import autotune, native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom
from distributed_tensorflow_amd.models.graph import same_padding
dev = torch.device("cuda")
N = 128
for (H, C, cfgs) in ((32, 16, [3, 7, 33]), (16, 32, [2, 15, 7, 32, 33, 34]), (8, 64, [5, 21, 14, 15, 32, 34])):
  pad = same_padding(H, 3, 1) + same_padding(H, 3, 1)
  g = ConvGeom(N, H, H, C, 3, 3, C, 1, pad)
  x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
  w = torch.randn(C * 9 * C + 4096, device=dev).to(torch.bfloat16)
  out = torch.empty(g.output_shape, dtype=torch.bfloat16, device=dev)
  stats = torch.zeros(32 * 2 * C, device=dev)
  for label, extra in (("plain", {}), ("+stats", dict(stats=stats.data_ptr(), stats_replicas=32))):
    row = []
    for cfg in cfgs:
      d = g.fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), cfg=cfg)
      d.update(extra)
      row.append("%d:%.1f" % (cfg, autotune._time_op("conv", d, iters=20) * 1e3))
    print("%dx%dx%d 3x3 %-7s %s" % (H, H, C, label, "  ".join(row)), flush=True)
# an empty-ish kernel for the launch floor
z = torch.zeros(64, device=dev)
print("memset 256 B: %.1f us" % (autotune._time_op("memset", dict(ptr=z.data_ptr(), nbytes=256), iters=20) * 1e3))
