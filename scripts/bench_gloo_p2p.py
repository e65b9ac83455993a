"""Point-to-point bandwidth of gloo on this host (two ranks, pinned host buffers): what the
host-staged parameter-server rehearsal can reach at best.

    python -m distributed_tensorflow_amd.launch --nproc 2 scripts/bench_gloo_p2p.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"]),
                        rank=rank, world_size=world)
pin = torch.cuda.is_available()
for mb in (2, 13, 26, 51, 102):
  n = mb * (1 << 20) // 2
  a = torch.empty(n, dtype=torch.bfloat16, pin_memory=pin)
  b = torch.empty(n, dtype=torch.bfloat16, pin_memory=pin)
  for mode in ("one-way", "duplex"):
    dist.barrier()
    t0 = time.time()
    for _ in range(5):
      if mode == "one-way":
        (dist.send(a, 1) if rank == 0 else dist.recv(a, 0))
      else:
        ws = [dist.isend(a, 1 - rank), dist.irecv(b, 1 - rank)]
        for w in ws:
          w.wait()
    dist.barrier()
    dt = (time.time() - t0) / 5
    if rank == 0:
      print("%4d MB %-8s %7.2f ms  %6.2f GB/s per direction" % (mb, mode, dt * 1e3, mb / 1024 / dt), flush=True)

# the same through several process groups (each has its own sockets and progress thread)
for ngroups in (2, 4):
  groups = [dist.new_group(ranks=[0, 1], backend="gloo") for _ in range(ngroups)]
  mb = 51
  n = mb * (1 << 20) // 2
  a = torch.empty(n, dtype=torch.bfloat16, pin_memory=pin)
  b = torch.empty(n, dtype=torch.bfloat16, pin_memory=pin)
  per = ngroups // 2
  chunk = -(-n // per)
  dist.barrier()
  t0 = time.time()
  for _ in range(5):
    ws = []
    for i in range(per):
      sl = slice(i * chunk, min(n, (i + 1) * chunk))
      ws.append(dist.isend(a[sl], 1 - rank, group=groups[2 * i + rank]))
      ws.append(dist.irecv(b[sl], 1 - rank, group=groups[2 * i + 1 - rank]))
    for w in ws:
      w.wait()
  dist.barrier()
  dt = (time.time() - t0) / 5
  if rank == 0:
    print("%4d MB duplex over %d groups %7.2f ms  %6.2f GB/s per direction" % (mb, ngroups, dt * 1e3, mb / 1024 / dt), flush=True)
