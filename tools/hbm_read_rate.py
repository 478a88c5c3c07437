"""What a plain streaming kernel reaches on this GPU (torch sum / copy of 2.8 GB): the scale for the HBM fractions of DESIGN.md section 4.
   gpurun -- 'python tools/hbm_read_rate.py'"""
import torch, time
x = torch.empty(700_000_000, dtype=torch.float32, device="cuda").normal_()
for f, name in ((lambda: x.sum(), "sum (read 2.8 GB)"), (lambda: x.abs().max(), "abs+max"), (lambda: torch.empty_like(x).copy_(x), "copy (r+w 5.6 GB)")):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    nbytes = x.numel()*4*(2 if "copy" in name else 1)
    print(f"{name}: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s")
