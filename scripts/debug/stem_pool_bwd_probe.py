"""Times the fused BatchNorm + ReLU + max-pool backward of the ResNet-50 stem (112x112x64, 3x3/2)
at batch 256.  DTF_BN_POOL_QUAD=0 selects the per-element kernels, the default the 2x2-patch
kernels; run once with each."""
import os
import torch
from distributed_tensorflow_amd.ops import native

dev = torch.device("cuda")
N, H, W, C, k, st = 256, 112, 112, 64, 3, 2
OH = OW = 56
y = torch.randn(N, H, W, C, device=dev).bfloat16()
dzp = torch.randn(N, OH, OW, C, device=dev).bfloat16()
idx = torch.randint(0, 10, (N, OH, OW, C), device=dev, dtype=torch.uint8)
saved = torch.zeros(4 * C, device=dev); saved[C:2 * C] = 1.0
gamma = torch.ones(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dy = torch.empty_like(y)
ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
d = dict(N=N, H=H, W=W, C=C, OH=OH, OW=OW, k=k, stride=st, pt=0, pl=0, idx=idx.data_ptr(),
         y=y.data_ptr(), gamma=gamma.data_ptr(), ws=ws.data_ptr(), save_mean=saved.data_ptr(),
         save_invstd=saved.data_ptr() + 4 * C, dzp=dzp.data_ptr(), dy=dy.data_ptr(),
         dgamma=dg.data_ptr(), dbeta=db.data_ptr())
for _ in range(3):
  native.run_op("bn_pool_bwd", d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
  native.run_op("bn_pool_bwd", d)
e1.record()
torch.cuda.synchronize()
print("DTF_BN_POOL_QUAD=%s: %.1f us per backward (reduce + finalize + apply)"
      % (os.environ.get("DTF_BN_POOL_QUAD", "1"), e0.elapsed_time(e1) * 50))
