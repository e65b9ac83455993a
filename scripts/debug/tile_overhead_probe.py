"""Per-tile overhead of the pipelined convolution kernel: a 1x1 layer with exactly one round of
blocks (512 tiles of 256x128, two per CU), reduction length varied.  time = overhead + KT * step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import autotune
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom
dev = torch.device("cuda")
N, H = 128, 32
for cfg, cout in ((17, 128), (30, 256), (23, 64)):
  for rounds in (1, 2, 4):
    row = []
    for cin in (32, 64, 128, 256, 512, 1024, 2048):
      g = ConvGeom(N * rounds, H, H, cin, 1, 1, cout, 1, (0, 0, 0, 0))
      x = torch.randn(N * rounds, H, H, cin, device=dev).to(torch.bfloat16)
      w = torch.randn(cin * cout + 4096, device=dev).to(torch.bfloat16)
      out = torch.empty(g.output_shape, dtype=torch.bfloat16, device=dev)
      d = g.fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), cfg=cfg)
      t = autotune._time_op("conv", d, iters=20) * 1e3
      row.append("K=%d:%.1f" % (cin, t))
      del x, out
    print("cfg %d Cout %d rounds %d  %s" % (cfg, cout, rounds, "  ".join(row)), flush=True)
