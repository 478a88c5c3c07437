"""Per-loss-term gradient comparison of the HIP LTS path against the CPU oracle (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F

from conftest import load_npz, rel_err
from esr_nerf_amd.config import lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.synthetic import slab_scene
from oracle import fine_path as fp
from oracle import lts_path as lp

mode = sys.argv[1] if len(sys.argv) > 1 else "lts"
z = {k: torch.from_numpy(np.asarray(v)) for k, v in load_npz(f"lts_g16_{mode}.npz").items()}
sd = {k: torch.from_numpy(v) for k, v in load_npz("lts_g16_params.npz").items()}
sc = slab_scene("g16", s_val=60.0, oblique=True)
cfg = lts_cfg("cuda:0", num_2ndrays=8, num_ltspts=12)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.xyz_min, sc.xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
m.train()
m.load_state_dict({k: v.cuda() for k, v in sd.items()})
m.pdra_mode = mode == "pdra"
tr = cfg.app.trainer
b = {k[3:]: v for k, v in z.items() if k.startswith("in/") and k != "in/s_val"}
bg = {k: v.cuda() for k, v in b.items()}
draws = {k[5:]: v for k, v in z.items() if k.startswith("draw/")}

ccfg = lts_cfg("cpu", num_2ndrays=8, num_ltspts=12)
c = fp.make_consts(ccfg.app.model, sc.xyz_min, sc.xyz_max, sc.xyz_min, sc.xyz_max, sc.mask_alpha_init,
                   sc.mask_density, sc.near, sc.num_voxels)

TERMS = {
    "base": lambda r, rgbs: fp.fine_loss(r, rgbs, True, tr.weight_linear, tr.weight_entropy_last)[0],
    "off": lambda r, rgbs: F.mse_loss(r["lin/pbr/off"], r["lin/pbr/off_hat"]),
    "off_pred_only": lambda r, rgbs: F.mse_loss(r["lin/pbr/off"], r["lin/pbr/off_hat"].detach()),
    "off_hat_only": lambda r, rgbs: F.mse_loss(r["lin/pbr/off"].detach(), r["lin/pbr/off_hat"]),
    "emo": lambda r, rgbs: F.mse_loss(r["lin/pbr/emo"], r["lin/pbr/emo_hat"]),
    "emo_hat_only": lambda r, rgbs: F.mse_loss(r["lin/pbr/emo"].detach(), r["lin/pbr/emo_hat"]),
    "normal": lambda r, rgbs: F.l1_loss(r["etc/normal"], r["etc/normal_eps"]),
    "normal_sq": lambda r, rgbs: ((r["etc/normal"] - r["etc/normal_eps"]) ** 2).sum(),
    "normal_lin": lambda r, rgbs: (r["etc/normal"] * torch.linspace(-1, 1, r["etc/normal"].numel(), device=r["etc/normal"].device).view_as(r["etc/normal"])).sum()
                                  + (r["etc/normal_eps"] * torch.linspace(2, -1, r["etc/normal"].numel(), device=r["etc/normal"].device).view_as(r["etc/normal"])).sum(),
    "emit": lambda r, rgbs: (r["etc/emit"] ** 2).mean() + (r["etc/emit_uncert"] ** 2).mean() + r["etc/emit_cert"].mean(),
    "brdf": lambda r, rgbs: (r["etc/brdf"] ** 2).mean(),
    "emit_eps": lambda r, rgbs: ((r["etc/emit"] - r["etc/emit_eps"]) ** 2).mean(),
    "brdf_eps": lambda r, rgbs: ((r["etc/brdf"] - 2 * r["etc/brdf_eps"]) ** 2).mean(),
}
for name, fn in TERMS.items():
    P = fp.params_from_state_dict(sd)
    res = lp.forward_training(P, c, b, 60.0, lp.Draws(**draws), tr.normal_eps, tr.emit_eps, 8, ccfg.app.model.lts_near,
                              pdra_mode=(mode == "pdra"))
    fn(res, b["rgbs"]).backward()
    m.zero_grad(set_to_none=True)
    rg = m(rays_o=bg["rays_o"], rays_d=bg["rays_d"], viewdirs=bg["viewdirs"], em_modes=bg["em_modes"],
           uncert_masks=bg["uncert_masks"], s_val=60.0, normal_eps=tr.normal_eps, emit_eps=tr.emit_eps,
           draws={k: v.cuda() for k, v in draws.items()})
    fn(rg, bg["rgbs"]).backward()
    errs = {}
    for k, p in m.named_parameters():
        go = P[k].grad if k in P else None
        if go is None and p.grad is None:
            continue
        if go is None:
            if float(p.grad.abs().max()) > 0:
                errs[k] = ("oracle none", float(p.grad.abs().max()))
            continue
        if p.grad is None:
            errs[k] = ("gpu none", float(go.abs().max()))
            continue
        e = rel_err(p.grad, go)
        if e > 2e-5:
            errs[k] = (e, float(go.abs().max()))
    if name == "normal":
        d = (rg["etc/normal"] - rg["etc/normal_eps"]).detach().cpu()
        do = (res["etc/normal"] - res["etc/normal_eps"]).detach()
        print("  sign flips:", int((torch.sign(d) != torch.sign(do)).sum()), "of", d.numel(), " |diff|<1e-6:", int((do.abs() < 1e-6).sum()))
    print(f"[{name}]", errs if errs else "all < 2e-5")
