import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29555")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from esr_nerf_amd.grad_sync import GridGradSync
n = 54525952
flat = torch.zeros(n, device="cuda")
idx = torch.randperm(n // 128, device="cuda")[:68890]
flat.view(-1, 128)[idx] = 1.0
s = GridGradSync(dist.group.WORLD)
small = torch.zeros(200000, device="cuda"); loss = torch.zeros(1, device="cuda")
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.ops.flags(flat, s._flags if s._flags is not None else torch.empty((n+127)//128, dtype=torch.uint8, device="cuda")) if False else None
    s.reduce(flat)
    t1 = time.perf_counter()
    dist.all_reduce(small); dist.all_reduce(loss)
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"reduce host {1e3*(t1-t0):.3f} ms, 2 small ARs host {1e3*(t2-t1):.3f} ms, drain {1e3*(t3-t2):.3f} ms, total {1e3*(t3-t0):.3f}")
dist.destroy_process_group()
