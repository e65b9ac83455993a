"""What a plain streaming kernel reaches on this GPU (yardstick for the BatchNorm sweeps):
device-to-device copies and a read-only reduction over tensors of the sizes ResNet-50 moves."""
import torch
dev = torch.device("cuda")
for mb in (51, 103, 205, 411, 1024, 4096):
  n = mb * (1 << 20) // 2
  a = torch.randn(n, device=dev).to(torch.bfloat16)
  b = torch.empty_like(a)
  for name, fn, touched in (("copy", lambda: b.copy_(a), 2 * a.numel() * 2),
                            ("read (sum)", lambda: a.sum(dtype=torch.float32), a.numel() * 2)):
    for _ in range(3):
      fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
      fn()
    t1.record()
    t1.synchronize()
    ms = t0.elapsed_time(t1) / 20
    print("%5d MB %-10s %8.1f us  %5.2f TB/s" % (mb, name, ms * 1e3, touched / ms / 1e9), flush=True)
