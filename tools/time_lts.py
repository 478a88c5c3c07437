"""Wall time + per-call HIP-event breakdown of one LTS training step (LtsStep, no autograd) at C4 size."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from esr_nerf_amd.config import lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import LtsStep

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s_val = float(sys.argv[2]) if len(sys.argv) > 2 else 220.0
sc = slab_scene("C2", s_val=s_val, n_rays=n_rays)
torch.manual_seed(0); np.random.seed(0)
cfg = lts_cfg("cuda:0")
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
m.train()
init_slab_model(m, sc, seed=1)
b = {k: v.cuda() for k, v in sc.batch.items()}
b["uncert_masks"] = torch.arange(n_rays, device="cuda") % 3 == 0
runner = LtsStep(m, cfg.app.trainer, stage=sys.argv[3] if len(sys.argv) > 3 else "lts")

def step():
    return runner.forward_loss_backward(b, s_val)[0]

for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.time()
K = 10
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.time() - t) / K
print(f"step {dt*1e3:.2f} ms  -> {n_rays/dt:.0f} rays/s   counts prim {m.engine.prim.counts} sec {m.engine.sec.counts}")
m.engine.enable_timing(True)
for _ in range(3):
    step()
tab = m.engine.timing_summary()
tot = sum(v[1] for v in tab.values()) / 3
print(f"sum of bracketed calls {tot:.2f} ms/step")
for k, (n, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {k:32s} {n//3:3d}x {ms/3:8.3f} ms")
