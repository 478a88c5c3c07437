"""Host time between consecutive C-ABI launches of one trainer step (where the Python side of a step goes, call by call):
   python tools/host_gaps.py [--config C5] [--steps 20]      (GPU box)
Every `_run` of the engine is stamped with time.perf_counter() on entry and exit; the table lists, per call in program
order, the host time since the previous call returned (= Python glue in front of it) and the time inside the call (ctypes +
hipLaunchKernel), averaged over the steps.  Event.synchronize waits are listed as their own rows."""
import argparse, contextlib, io, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5")
ap.add_argument("--steps", type=int, default=20)
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
eng = model.engine
log = []
orig_run = eng._run
def stamped(name, fn, *args):
    t0 = time.perf_counter()
    orig_run(name, fn, *args)
    log.append((name, t0, time.perf_counter()))
eng._run = stamped
_sync = torch.cuda.Event.synchronize
def timed_sync(self):
    t0 = time.perf_counter()
    _sync(self)
    log.append(("<Event.synchronize>", t0, time.perf_counter()))
torch.cuda.Event.synchronize = timed_sync
import gc; gc.collect(); gc.freeze(); gc.disable()
rows = {}
order = []
for it in range(a.steps):
    log.clear()
    t_prev = time.perf_counter()
    run()
    t_end = time.perf_counter()
    seen = {}
    for name, t0, t1 in log:
        k = seen.get(name, 0); seen[name] = k + 1
        key = f"{name}#{k}" if k else name
        if key not in rows:
            rows[key] = [0.0, 0.0, 0]; order.append(key)
        rows[key][0] += t0 - t_prev; rows[key][1] += t1 - t0; rows[key][2] += 1
        t_prev = t1
    rows.setdefault("<return>", [0.0, 0.0, 0]); rows["<return>"][0] += t_end - t_prev; rows["<return>"][2] += 1
    if "<return>" not in order: order.append("<return>")
torch.cuda.synchronize()
tot_gap = tot_in = 0.0
print(f"{a.config}: host time per call, us (glue in front of it | inside it), mean of {a.steps} steps")
for k in order:
    g, i, n = rows[k]
    if "synchronize" not in k:
        tot_gap += g / a.steps; tot_in += i / a.steps
    print(f"  {g / n * 1e6:7.1f} | {i / n * 1e6:7.1f}   {k}")
print(f"sum per step: glue {tot_gap * 1e3:.3f} ms, inside calls {tot_in * 1e3:.3f} ms (waits not counted in `inside`)")
