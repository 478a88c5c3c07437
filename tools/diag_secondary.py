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
# ---- where the survivors of one surface point's 256 rays lie, relative to the point (what a per-point accumulation window in
# feat_bwd could catch): cells from the ray origin + direction * (near + step * stepdist)
cap = {}
orig_march = eng._march
def spy(P, scene_, o, d, *a, **k):
    if P is eng.sec:
        cap["o"], cap["d"], cap["near"], cap["stepdist"] = o.detach().clone(), d.detach().clone(), float(scene_.near_), float(scene_.stepdist)
    return orig_march(P, scene_, o, d, *a, **k)
eng._march = spy
step.forward_loss_backward(batch, 220.0)
torch.cuda.synchronize()
eng._march = orig_march
rr = P2.bufs["rec_ray"].cpu().numpy(); rs = P2.bufs["rec_step"].cpu().numpy()
ok = rr >= 0
o, d = cap["o"].cpu().numpy(), cap["d"].cpu().numpy()
d = d / np.linalg.norm(d, axis=1, keepdims=True)
pos = o[rr[ok]] + d[rr[ok]] * (cap["near"] + rs[ok][:, None] * cap["stepdist"])
lo, hi = np.array(scene.xyz_min), np.array(scene.xyz_max)
dims = np.array([int(v) for v in model.world_size.tolist()])
cell = np.floor((pos - lo) / (hi - lo) * (dims - 1)).astype(np.int64)
pcell = np.floor((o[rr[ok]] - lo) / (hi - lo) * (dims - 1)).astype(np.int64)
dist = np.abs(cell - pcell).max(1)
print("near %.4f stepdist %.4f; survivors: distance (cells, max-norm) from their surface point: p25 %d p50 %d p75 %d p90 %d" % (
    cap["near"], cap["stepdist"], *np.percentile(dist, [25, 50, 75, 90])))
for half in (4, 6, 8, 12):
    print("  share of survivors within +-%d cells of their point: %.3f" % (half, (dist <= half).mean()))
pt = rr[ok] // 256
key = (cell[:, 0] * dims[1] + cell[:, 1]) * dims[2] + cell[:, 2]
per = [(np.sum(pt == p), len(np.unique(key[pt == p]))) for p in np.unique(pt)[:100]]
print("per surface point: samples mean %.0f, distinct base cells mean %.0f (ratio %.1f)" % (
    np.mean([a for a, b in per]), np.mean([b for a, b in per]), np.sum([a for a, b in per]) / max(1, np.sum([b for a, b in per]))))
