"""How long a pause (idle device) does it take for the step's ramp to come back?  (GPU box)  python tools/debug/pause_ramp.py"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from esr_nerf_amd.config import fine_cfg
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import FineStep
from esr_nerf_amd.voxurff import VoxurfF
sc = slab_scene("C2", s_val=20.0)
torch.manual_seed(0); np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = VoxurfF(fine_cfg("cuda:0"), sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc)
m.train()
b = {k: v.cuda() for k, v in sc.batch.items()}
step = FineStep(m)


def run(k):
    out = []
    for _ in range(k):
        t = time.perf_counter()
        step.forward_loss_backward(b, 20.0)
        torch.cuda.synchronize()
        out.append(round(1e3 * (time.perf_counter() - t), 3))
    return out


print("cold start      :", run(40)[:12], "...")
for pause in (0.0005, 0.005, 0.05, 0.5):
    time.sleep(pause)
    print(f"after {pause * 1e3:6.1f} ms idle:", run(12))
    run(20)
x = torch.randn(8192, 8192, device="cuda")
for mb in (64, 512, 4096):
    run(20)
    y = torch.empty(mb * 2**20 // 4, device="cuda")
    y.fill_(1.0); y.mul_(2.0)                              # streams 3 x mb MB through the memory system
    torch.cuda.synchronize()
    print(f"after streaming {mb:4d} MB :", run(12))
    del y
