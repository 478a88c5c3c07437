"""Is an LtsStep a deterministic function of its inputs and draws (up to float-atomic order, ~1e-6)?  Two processes share the GPU
(as in tests/test_gpu_dp.py: time slicing changes the streams' relative timing); each repeats the SAME step with the SAME draws and
compares every result, the loss and every gradient with the first repetition.  A cross-stream race shows as a rare large difference.
   python tools/debug/lts_repeat.py [serial]     ("serial": overlap_wgrad off = one stream)
   (run two copies side by side:  python tools/debug/lts_repeat.py & python tools/debug/lts_repeat.py; wait)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esr_nerf_amd.config import lts_cfg                                   # noqa: E402
from esr_nerf_amd.esrnerf import ESRNeRF                                  # noqa: E402
from esr_nerf_amd.synthetic import init_slab_model, slab_scene            # noqa: E402
from esr_nerf_amd.trainer import LtsStep                                  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
s_val = 60.0
sc = slab_scene("small", s_val=s_val, oblique=True, n_rays=192, seed=2)
torch.manual_seed(0); np.random.seed(0)
cfg = lts_cfg("cuda:0", num_2ndrays=16, num_ltspts=25)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc, seed=3)
with torch.no_grad():
    m.brdf.grid.normal_(0.0, 0.3, generator=torch.Generator(device="cuda").manual_seed(5))
m.train()
eng = m.engine
if mode == "serial":
    eng.overlap_wgrad = False
elif mode == "noeps":
    eng.eps_stream = False
elif mode == "noscatter":
    eng.scatter_streamed = set()
elif mode == "nowgradearly":
    eng.wgrad_early = set()
_stash = {}
_fwd = eng.lts_forward
def _spy(*a, **k):
    ctx, out = _fwd(*a, **k)
    _stash["ctx"] = ctx
    return ctx, out
eng.lts_forward = _spy
b = {k: v.cuda() for k, v in sc.batch.items()}
b["uncert_masks"] = (torch.arange(192, device="cuda") % 3 == 0)
for stage in ("lts", "pdra"):
    m.pdra_mode = stage == "pdra"
    step = LtsStep(m, cfg.app.trainer, stage=stage)
    step.forward_loss_backward(b, s_val)
    m3 = m.last_counts["m3"]
    g = torch.Generator().manual_seed(1)
    draws = dict(idx=torch.randperm(m3, generator=g)[:25].cuda(), dirs=torch.randn(25, 17, 3, generator=g).cuda(),
                 noise_normal=torch.randn(m3, 3, generator=g).cuda(), noise_emit=torch.randn(m3, 3, generator=g).cuda())
    ref, bad = None, 0
    seeded = os.environ.get("ESR_REPEAT_SEEDED", "0") == "1"       # the engine draws for itself (host point draw, eps stream)
    for r in range(reps):
        if seeded:
            if r % 2:                                # a different step in between (the pinned ring alternates)
                torch.manual_seed(7); np.random.seed(7)
                step.forward_loss_backward(b, s_val)
            torch.manual_seed(100); np.random.seed(100)
            loss, G, out = step.forward_loss_backward(b, s_val)
        else:
            loss, G, out = step.forward_loss_backward(b, s_val, draws=draws)
        torch.cuda.synchronize()
        cur = {"loss": loss.clone()}
        cur.update({"out/" + k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
        cur.update({"grad/" + k: v.clone() for k, v in G.items()})
        if seeded:
            c = _stash["ctx"]
            cur.update({"ctx/perm": c.perm.clone(), "ctx/jp": c.jp.clone(), "ctx/eg": c.t["eg"].clone(), "ctx/noise_n": c.t["noise_n"].clone(),
                        "ctx/pts_e": c.t["pts_e"].clone(), "ctx/pts_all": c.t["pts_all"].clone()})
            cur.update({"draw/" + k: v.clone().cuda() for k, v in eng.last_draws.items()})
        if ref is None:
            ref = cur
            continue
        worst = []
        for k, v in cur.items():
            if k == "ctx/eg":          # padding slots are not defined: compare the slots of the reference order
                v, w = v[cur["ctx/perm"]], ref[k][ref["ctx/perm"]]
            else:
                w = ref[k]
            d = float((v.double() - w.double()).abs().max() / w.double().abs().max().clamp_min(1e-30))
            if d > 2e-5:
                rows = (v.double() - w.double()).abs().reshape(v.shape[0], -1).amax(1) > 2e-5 * float(w.double().abs().max()) if v.dim() else None
                nrow = int(rows.sum()) if rows is not None else 1
                first = rows.nonzero().flatten()[:6].tolist() if rows is not None else []
                if k in ("ctx/eg", "out/etc/normal_eps", "out/etc/emit_eps") and first:
                    i0 = first[0]
                    print(f"   {k} rows {i0 - 1}..{i0 + 2}: now {v[max(i0 - 1, 0):i0 + 3].tolist()} first {w[max(i0 - 1, 0):i0 + 3].tolist()}", flush=True)
                    pm = cur["ctx/perm"]
                    rows_all = rows.nonzero().flatten()
                    print(f"   compact slots {pm[rows_all].tolist()}  rays {eng.prim.bufs['rec_ray'][pm[rows_all]].tolist()} steps {eng.prim.bufs['rec_step'][pm[rows_all]].tolist()}", flush=True)
                worst.append((k + f"[{nrow} of {v.shape[0] if v.dim() else 1} rows, first {first}]", d))
        if worst:
            bad += 1
            print(f"[pid {os.getpid()} {mode} {stage}] repetition {r}: " + ", ".join(f"{k} {d:.1e}" for k, d in sorted(worst, key=lambda kv: -kv[1])[:8]), flush=True)
    print(f"[pid {os.getpid()} {mode} {stage}] {bad} of {reps - 1} repetitions differ from the first by more than 2e-5", flush=True)
