"""Tile quantisation probe: the 14x14x256 3x3 layer at batch sizes that fill 392 .. 784 block
slots with 256x128 tiles (cfg 17) and 196 .. 392 with 256x256 tiles (cfg 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import autotune
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom
from distributed_tensorflow_amd.models.graph import same_padding
dev = torch.device("cuda")
H, C = 14, 256
pad = same_padding(H, 3, 1) + same_padding(H, 3, 1)
for N in (192, 224, 256, 288, 320, 334, 352, 384, 448, 512):
  g = ConvGeom(N, H, H, C, 3, 3, C, 1, pad)
  x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
  w = torch.randn(C * 9 * C + 4096, device=dev).to(torch.bfloat16)
  out = torch.empty(g.output_shape, dtype=torch.bfloat16, device=dev)
  row = []
  for cfg in (17, 30, 6, 0):
    d = g.fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), cfg=cfg)
    t = autotune._time_op("conv", d, iters=20) * 1e3
    row.append("cfg%d %.1f us (%.0f TF/s)" % (cfg, t, 2 * N * H * H * C * 9 * C / t / 1e6))
  tiles = -(-N * H * H // 256)
  print("N=%3d  %4d row tiles  %s" % (N, tiles, "   ".join(row)), flush=True)
