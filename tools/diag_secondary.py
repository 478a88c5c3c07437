"""Survivor statistics of the C4 secondary pass (which steps of the 25.6 k secondary rays survive, how tiles mix rays):
   python tools/diag_secondary.py   (GPU box)"""
import os, sys, io, contextlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
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
model.train()
with torch.no_grad():
    model.brdf.grid.normal_(0.0, 0.1)
batch = {k: v.to(dev) for k, v in scene.batch.items()}
batch["uncert_masks"] = (torch.arange(scene.n_rays, device=dev) % 3 == 0)
step = LtsStep(model, cfg.app.trainer, stage="lts")
for _ in range(2):
    step.forward_loss_backward(batch, 220.0)
torch.cuda.synchronize()
eng = model.lts_engine if hasattr(model, "lts_engine") else model.engine
P2 = eng.sec
rr = P2.bufs["rec_ray"].cpu().numpy(); rs = P2.bufs["rec_step"].cpu().numpy()
ok = rr >= 0
T = (np.nonzero(ok)[0].max() + 32) // 32
rr, rs, ok = rr[:T * 32], rs[:T * 32], ok[:T * 32]
print("tiles", T, "samples", ok.sum(), "rays with survivors", len(np.unique(rr[ok])))
cnt = np.bincount(rr[ok]); cnt = cnt[cnt > 0]
print("survivors per ray: mean %.1f  median %d  p90 %d  max %d" % (cnt.mean(), np.median(cnt), np.percentile(cnt, 90), cnt.max()))
print("step index of survivors: p10 %d p50 %d p90 %d max %d; share with step < 8: %.2f, < 16: %.2f" % (
    np.percentile(rs[ok], 10), np.median(rs[ok]), np.percentile(rs[ok], 90), rs[ok].max(), (rs[ok] < 8).mean(), (rs[ok] < 16).mean()))
rpt = [len(np.unique(rr[t * 32:(t + 1) * 32][ok[t * 32:(t + 1) * 32]])) for t in range(T)]
print("rays per tile: mean %.2f  hist" % np.mean(rpt), np.bincount(rpt)[:12])
ppt = [len(np.unique(rr[t * 32:(t + 1) * 32][ok[t * 32:(t + 1) * 32]] // 256)) for t in range(T)]
print("surface points per tile: hist", np.bincount(ppt)[:6])
