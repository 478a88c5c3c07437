"""Host-side cost of one trainer step (python + ctypes + torch launch overhead): cProfile over N steps without device syncs
between them.   python tools/host_profile.py [--config C5] [--steps 20]      (GPU box)"""
import argparse, cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--top", type=int, default=35)
a = ap.parse_args()
from esr_nerf_amd.config import fine_cfg, lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.voxurff import VoxurfF
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import FineStep, LtsStep
dev = "cuda:0"
fine = a.config in ("C2", "C3")
scene = slab_scene("C4" if a.config == "C5" else a.config, s_val=20.0 if fine else 220.0, seed=0)
torch.manual_seed(0); np.random.seed(0)
cfg = fine_cfg(dev) if fine else lts_cfg(dev)
with contextlib.redirect_stdout(io.StringIO()):
    model = (VoxurfF if fine else ESRNeRF)(cfg, scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min,
                                         scene.mask_xyz_max, scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
init_slab_model(model, scene)
model.mlp_dtype = "bf16" if a.config in ("C5", "C3") else "f32"
model.train()
batch = {k: v.to(dev) for k, v in scene.batch.items()}
if fine:
    step = FineStep(model)
    run = lambda: step.forward_loss_backward(batch, 20.0)
else:
    model.pdra_mode = a.config == "C5"
    with torch.no_grad():
        model.brdf.grid.normal_(0.0, 0.1)
    batch["uncert_masks"] = (torch.arange(scene.n_rays, device=dev) % 3 == 0)
    step = LtsStep(model, cfg.app.trainer, stage="pdra" if a.config == "C5" else "lts")
    run = lambda: step.forward_loss_backward(batch, 220.0)
for _ in range(5):
    run()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze(); gc.disable()
# time the host spends WAITING for the device inside a step (the plan read-backs): the rest of `host returned after` is work
_wait = [0.0]
_sync = torch.cuda.Event.synchronize
def _timed_sync(self):
    t = time.perf_counter()
    _sync(self)
    _wait[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = _timed_sync
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
torch.cuda.Event.synchronize = _sync
print(f"{a.config}: {t_all / a.steps * 1e3:.3f} ms per step, host returned after {t_host / a.steps * 1e3:.3f} ms per step, "
      f"of which {_wait[0] / a.steps * 1e3:.3f} ms waiting in Event.synchronize (host work: {(t_host - _wait[0]) / a.steps * 1e3:.3f} ms)")
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    run()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(a.top)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:9000])
