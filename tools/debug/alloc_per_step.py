"""Per-step host wall time and device allocations (hipMalloc calls of torch's caching allocator) of the C5 / C4 step:
   python tools/debug/alloc_per_step.py [--config C5] [--steps 40]      (GPU box)"""
import argparse, contextlib, io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5", choices=["C4", "C5"])
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
from esr_nerf_amd.config import lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import LtsStep
dev = "cuda:0"
scene = slab_scene("C4", s_val=220.0, seed=0)
torch.manual_seed(0); np.random.seed(0)
cfg = lts_cfg(dev)
with contextlib.redirect_stdout(io.StringIO()):
    model = ESRNeRF(cfg, scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min, scene.mask_xyz_max,
                    scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
init_slab_model(model, scene)
model.mlp_dtype = "bf16" if a.config == "C5" else "f32"
model.train()
batch = {k: v.to(dev) for k, v in scene.batch.items()}
model.pdra_mode = a.config == "C5"
with torch.no_grad():
    model.brdf.grid.normal_(0.0, 0.1)
batch["uncert_masks"] = (torch.arange(scene.n_rays, device=dev) % 3 == 0)
step = LtsStep(model, cfg.app.trainer, stage="pdra" if a.config == "C5" else "lts")
stats = lambda: torch.cuda.memory_stats(dev)
line = []
for it in range(a.steps):
    n0, f0 = stats()["num_device_alloc"], stats()["num_device_free"]
    t0 = time.perf_counter()
    step.forward_loss_backward(batch, 220.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s = stats()
    line.append(f"step {it:2d}: host {1e3 * (t1 - t0):6.2f} ms, to idle {1e3 * (t2 - t0):6.2f} ms, hipMalloc {s['num_device_alloc'] - n0}, hipFree {s['num_device_free'] - f0}, "
                f"reserved {s['reserved_bytes.all.current'] / 2**30:.2f} GiB, secondary tiles {model.engine.sec.tiles_all}")
print("\n".join(line))
