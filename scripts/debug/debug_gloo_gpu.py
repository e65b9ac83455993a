"""Two gloo ranks exchanging device tensors on one GPU (start with the launcher, 2 ranks)."""
import os, sys, torch, torch.distributed as dist, datetime, time
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
print("rank",rank,"start",flush=True)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s"%os.environ["MASTER_PORT"], rank=rank, world_size=world, timeout=datetime.timedelta(seconds=40))
print("rank",rank,"pg ok",flush=True)
t=torch.ones(1000,device="cuda")*(rank+1)
w=dist.all_reduce(t, async_op=True)
print("rank",rank,"issued",flush=True)
w.wait()
torch.cuda.synchronize()
print("rank",rank,"allreduce",t[0].item(),flush=True)
b=torch.zeros(10,device="cuda")+rank
dist.broadcast(b,0)
print("rank",rank,"bcast",b[0].item(),flush=True)
dist.barrier()
print("rank",rank,"done",flush=True)
