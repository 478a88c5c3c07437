"""Generate tests/golden/*.npz from the IMPORTED reference -- build container only.

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.gen_golden
It imports /root/reference (oracle/ref_import.py), runs the reference's own
``VoxurfF.forward_training`` + the ``Fine.learn`` loss arithmetic + backward on
small slab scenes, and stores inputs, parameters, outputs, gradients and the
inputs/outputs of every native-op call as data.  The fixtures are what pins
(a) the C oracle, (b) oracle/fine_path.py and (c) the HIP path; the reference
source itself never leaves this container.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from esr_nerf_amd.config import fine_cfg
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from oracle import native, ref_import

# ESR_GOLDEN_OUT: write somewhere else (tests/test_golden_regen.py regenerates into a scratch directory and diffs)
OUT = os.environ.get("ESR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                       "tests", "golden")

CASES = {
    # name: (scene kwargs, s_val)
    "fine_g16_axis": (dict(name="g16", oblique=False), 20.0),
    "fine_g16_oblique": (dict(name="g16", oblique=True), 60.0),
    # pruning mask cache (synthetic.prune_mask): M0 > M1 > M2 > M3, non-contiguous survivors
    "fine_g16_prune_axis": (dict(name="g16", oblique=False, mask="prune"), 20.0),
    "fine_g16_prune_oblique": (dict(name="g16", oblique=True, mask="prune"), 60.0),
    # data.white_bg = False (the dtu configs, cfg/data/dtu.yaml): the loss adds no background term
    "fine_g16_prune_oblique_nobg": (dict(name="g16", oblique=True, mask="prune"), 60.0),
    # cfg neus_alpha: "grad" (functions.py:45-69): per-sample extrapolation with the finite-difference gradient
    "fine_g16_prune_oblique_gradalpha": (dict(name="g16", oblique=True, mask="prune"), 60.0),
}


def _sfx(mask):
    return "" if mask == "full" else "_" + mask


def reference_loss(ns, results, rgbs, cfg):
    """Arithmetic of app/fine/fine.py:355-382 evaluated with the reference's own
    apply_gamma_curve (utils2/image.py:14-26)."""
    tr = cfg.app.trainer
    white_bg = results["etc/white_bg"] * (1.0 if cfg.data.white_bg else 0.0)
    srgb = (results["srgb/rgb"] + white_bg).clamp(min=0.0, max=1.0)
    lin = (results["lin/rgb"] + white_bg).clamp(min=0.0)
    loss = F.mse_loss(srgb, rgbs)
    lin_loss = F.mse_loss(
        ns.image.apply_gamma_curve(torch.where(rgbs >= 1, lin.clamp(max=1.0), lin)), rgbs)
    loss = loss + tr.weight_linear * lin_loss
    pout = results["etc/alphainv_cum"][..., -1].clamp(1e-6, 1 - 1e-6)
    ent = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
    return loss + tr.weight_entropy_last * ent


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load()
    cfg = fine_cfg("cpu")

    models = {}
    for mask in ("full", "prune", "prune/grad"):
        torch.manual_seed(0)
        np.random.seed(0)
        base = slab_scene("g16", mask=mask.split("/")[0])
        mcfg = cfg if "/" not in mask else fine_cfg("cpu", neus_alpha="grad")
        model = ns.VoxurfF(mcfg, base.near, base.far, base.xyz_min, base.xyz_max, base.mask_xyz_min,
                           base.mask_xyz_max, base.mask_alpha_init, base.mask_density, base.s_val,
                           base.num_voxels)
        init_slab_model(model, base)
        model.train()
        models[mask] = model
    sd = {k: v.detach().clone() for k, v in models["full"].state_dict().items()}
    for mk in ("prune", "prune/grad"):
        for k, v in models[mk].state_dict().items():
            assert torch.equal(v, sd[k]), k        # mask cache and alpha mode are not parameters: one parameter file
    model = models["full"]
    np.savez_compressed(
        os.path.join(OUT, "fine_g16_params.npz"),
        **{k: v.numpy() for k, v in sd.items()},
        __world_size=model.world_size.numpy(), __voxel_size=model.voxel_size.numpy(),
    )

    # record native-op traffic
    rec = {}
    ru = ns.functions.render_utils_cuda
    real_sample, real_a2w, real_a2w_b = ru.sample_pts_on_rays, ru.alpha2weight, ru.alpha2weight_backward

    def rec_sample(*a):
        out = real_sample(*a)
        rec["sample_in"] = [torch.as_tensor(x).detach().clone() for x in a]
        rec["sample_out"] = [x.detach().clone() for x in out]
        return out

    def rec_a2w(alpha, ray_id, n):
        out = real_a2w(alpha, ray_id, n)
        rec["a2w_in"] = [alpha.detach().clone(), ray_id.clone(), torch.tensor(n)]
        rec["a2w_out"] = [x.detach().clone() for x in out]
        return out

    def rec_a2w_b(*a):
        out = real_a2w_b(*a)
        rec["a2wb_grads"] = [a[7].detach().clone(), a[8].detach().clone()]
        rec["a2wb_out"] = out.detach().clone()
        return out

    ru.sample_pts_on_rays, ru.alpha2weight, ru.alpha2weight_backward = rec_sample, rec_a2w, rec_a2w_b

    for case, (skw, s_val) in CASES.items():
        sc = slab_scene(s_val=s_val, **skw)
        b = sc.batch
        model = models[skw.get("mask", "full") + ("/grad" if case.endswith("_gradalpha") else "")]
        model.zero_grad(set_to_none=True)
        rec.clear()
        hook = model.mask_cache.register_forward_hook(lambda mod, a, out: rec.__setitem__("mask_keep", out.clone()))
        res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"],
                    em_modes=b["em_modes"], s_val=s_val)
        hook.remove()
        res_raw = {k: v.detach().clone() for k, v in res.items()}
        for v in res.values():
            v.retain_grad()
        cfg_case = fine_cfg("cpu")
        cfg_case.data.white_bg = not case.endswith("_nobg")
        loss = reference_loss(ns, dict(res), b["rgbs"], cfg_case)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        out = {"in/white_bg": np.bool_(cfg_case.data.white_bg)}
        for k, v in b.items():
            out["in/" + k] = v.numpy()
        out["in/s_val"] = np.float32(s_val)
        for k, v in res_raw.items():
            out["out/" + k] = v.numpy()
            out["dout/" + k] = (res[k].grad if res[k].grad is not None else torch.zeros_like(res[k])).numpy()
        out["loss"] = loss.detach().numpy()
        for k, v in grads.items():
            out["grad/" + k] = v.numpy()
        names = ["ray_pts", "mask_outbbox", "ray_id", "step_id", "N_steps", "t_min", "t_max"]
        out["native/sample/stepdist"] = np.float32(float(rec["sample_in"][6]))
        out["native/sample/near"] = np.float32(float(rec["sample_in"][4]))
        for n_, t in zip(names, rec["sample_out"]):
            out["native/sample/" + n_] = t.numpy()
        out["native/a2w/alpha"] = rec["a2w_in"][0].numpy()
        out["native/a2w/ray_id"] = rec["a2w_in"][1].numpy()
        for n_, t in zip(["weight", "T", "alphainv_last", "i_start", "i_end"], rec["a2w_out"]):
            out["native/a2w/" + n_] = t.numpy()
        out["native/a2wb/grad_weights"] = rec["a2wb_grads"][0].numpy()
        out["native/a2wb/grad_last"] = rec["a2wb_grads"][1].numpy()
        out["native/a2wb/grad"] = rec["a2wb_out"].numpy()
        out["native/mask_keep"] = rec["mask_keep"].numpy()          # MaskCache.forward output over the M0 in-box samples
        np.savez_compressed(os.path.join(OUT, case + ".npz"), **out)
        print(case, "loss", float(loss), "M0", len(rec["sample_out"][0]), "M1", int(rec["mask_keep"].sum()),
              "M2", len(rec["a2w_in"][0]),
              {k: tuple(v.shape) for k, v in res_raw.items()})

    ru.sample_pts_on_rays, ru.alpha2weight, ru.alpha2weight_backward = real_sample, real_a2w, real_a2w_b
    gen_host(ns)
    for mask in ("full", "prune"):
        gen_lts(ns, mask)            # the "full" variant writes lts_g16_params.npz, which every later variant checks
        gen_finetune(ns, mask)
        gen_coarse(ns, mask)
        gen_eval(ns, mask)
        gen_lts_evals(ns, mask)
        gen_coarse_eval(ns, mask)
        gen_lts_eval(ns, mask)
    gen_lts(ns, "prune", "fib")
    gen_coarse(ns, "prune", neus_alpha="grad")
    gen_lts(ns, "prune", neus_alpha="grad")


def gen_host(ns):
    """Reference-pinned fixtures for the rows around the renderer: the imported
    app/utils/optimizer.py (Adam incl. per-voxel lr and CosineLR decay, 10 steps), and the imported VoxurfF's
    density_total_variation (+ autograd gradient), sdf_total_variation_add_grad, scale_volume_grid and
    filter_training_rays_in_maskcache_sampling in both sdf_random_init branches."""
    from esr_nerf_amd.config import AttrDict
    opt_mod = ns.optimizer
    out = {}
    # ---- CosineLR: factor sequences for the trainers' settings and the branches of cosine_lr_func
    sched = {"fine": dict(n_iters=200, warm_up_iters=20, warm_up_min_ratio=0.1, const_warm_up=False, cos_min_ratio=0.05),
             "const": dict(n_iters=120, warm_up_iters=30, warm_up_min_ratio=0.3, const_warm_up=True, cos_min_ratio=0.2),
             "allwarm": dict(n_iters=64, warm_up_iters=-1, warm_up_min_ratio=0.5, const_warm_up=False, cos_min_ratio=0.0)}
    for name, tr in sched.items():
        for start in (0, 37):
            c = opt_mod.CosineLR(AttrDict(app=dict(trainer=tr)), cur_step=start)
            out[f"cos/{name}/{start}"] = np.array([c.decay_factor for _ in range(tr["n_iters"] - start)], np.float64)
        out[f"cos/{name}/cfg"] = np.array([tr["n_iters"], tr["warm_up_iters"], tr["warm_up_min_ratio"],
                                           float(tr["const_warm_up"]), tr["cos_min_ratio"]], np.float64)
    # ---- Adam: three groups, the first (a 1-channel grid) with a per-voxel lr; lr decayed by CosineLR every step
    g = torch.Generator().manual_seed(31)
    shapes = {"grid1": (1, 1, 6, 5, 4), "grid6": (1, 6, 6, 5, 4), "w": (7, 5)}
    lrs = {"grid1": 0.1, "grid6": 0.05, "w": 0.003}

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for k, shp in shapes.items():
                setattr(self, k, torch.nn.Parameter(torch.randn(shp, generator=torch.Generator().manual_seed(len(k))) * 0.3))

    torch.manual_seed(0)
    m = Holder()
    for k in shapes:
        out[f"adam/p0/{k}"] = getattr(m, k).detach().numpy().copy()
    opt = opt_mod.create_optimizer_or_freeze_model(m, **lrs)
    count = torch.randint(0, 9, shapes["grid1"], generator=g)
    opt.set_pervoxel_lr(count)
    out["adam/count"] = count.numpy()
    cos = opt_mod.CosineLR(AttrDict(app=dict(trainer=sched["fine"])), cur_step=0)
    for step in range(10):
        for k in shapes:
            gr = torch.randn(shapes[k], generator=g) * (0.5 if step % 3 else 2.0)
            if k == "grid6" and step == 4:
                gr.zero_()                                           # an all-zero gradient step
            getattr(m, k).grad = gr
            out[f"adam/g{step}/{k}"] = gr.numpy().copy()
        opt.step()
        f = cos.decay_factor
        for pg in opt.param_groups:
            pg["lr"] = pg["lr"] * f
        for k in shapes:
            out[f"adam/p{step + 1}/{k}"] = getattr(m, k).detach().numpy().copy()
    for k in shapes:
        st = opt.state[getattr(m, k)]
        out[f"adam/m/{k}"], out[f"adam/v/{k}"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
    out["adam/lr_final"] = np.array([pg["lr"] for pg in opt.param_groups], np.float64)
    np.savez_compressed(os.path.join(OUT, "host_optimizer.npz"), **out)
    print("host optimizer fixture:", len(out), "arrays")

    # ---- dense-grid rows on a small prune-mask scene
    cfg = fine_cfg("cpu")
    sc = slab_scene("g16", mask="prune")
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.VoxurfF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
    init_slab_model(model, sc)
    with torch.no_grad():
        model.sdf.grid.add_(torch.randn(model.sdf.grid.shape, generator=torch.Generator().manual_seed(4)) * 0.02)
    model.set_nonempty_mask()
    model.train()
    o = {"sdf": model.sdf.grid.detach().numpy().copy(), "nonempty_mask": model.nonempty_mask.numpy().copy()}
    # density_total_variation (fine.py:384-393 weights): needs self.gradient as forward_training leaves it
    model.gradient = model.neus_sdf_gradient()
    for key, kw in (("tv_sdf", dict(sdf_tv=0.1)), ("tv_smooth", dict(smooth_grad_tv=0.05)), ("tv_both", dict(sdf_tv=0.1, smooth_grad_tv=0.05))):
        model.zero_grad(set_to_none=True)
        model.gradient = model.neus_sdf_gradient()
        tv = model.density_total_variation(**kw)
        tv.backward()
        o[key + "/value"] = tv.detach().numpy()
        o[key + "/grad_sdf"] = model.sdf.grid.grad.numpy().copy()
    # sdf_total_variation_add_grad (fine.py:397-401): dense and sparse mode on a half-empty gradient
    for dense in (True, False):
        gr = torch.randn(model.sdf.grid.shape, generator=torch.Generator().manual_seed(6)) * 1e-3
        gr[..., ::2, :] = 0.0
        model.sdf.grid.grad = gr.clone()
        model.sdf_total_variation_add_grad(0.1 / 3, dense)
        o[f"tv_add_grad/{int(dense)}/in"] = gr.numpy()
        o[f"tv_add_grad/{int(dense)}/out"] = model.sdf.grid.grad.numpy().copy()
    # filter_training_rays_in_maskcache_sampling, both branches, on rays that partly miss the box / the mask
    gg = torch.Generator().manual_seed(12)
    n = 300
    ro = torch.cat([(torch.rand(n, 2, generator=gg) * 2 - 1) * 1.3, torch.full((n, 1), 2.0)], -1)
    rd = torch.cat([(torch.rand(n, 2, generator=gg) * 2 - 1) * 0.4, -torch.ones(n, 1)], -1) * (0.5 + torch.rand(n, 1, generator=gg))
    rd[::19, 1] = 0.0
    o["filter/rays_o"], o["filter/rays_d"] = ro.numpy(), rd.numpy()
    far0 = model.far
    model.far = 2.6          # cuts the t-range of the slow rays: only the random-init sampler clamps to far (voxurff.py:520-525)
    o["filter/far"] = np.float32(model.far)
    for rnd in (False, True):
        model.sdf_random_init = rnd
        o[f"filter/keep/{int(rnd)}"] = model.filter_training_rays_in_maskcache_sampling(ro, rd, 128).numpy()
    model.sdf_random_init = False
    model.far = far0
    # scale_volume_grid: [32,32,8] -> the resolution of 4.096x the voxels (fine.yaml pg_scale)
    model.zero_grad(set_to_none=True)
    o["scale/off_color_in"] = model.off_color.grid.detach().numpy().copy()
    nv = int(sc.num_voxels * 4.096)
    model.scale_volume_grid(nv)
    o["scale/num_voxels"] = np.int64(nv)
    o["scale/world_size"] = model.world_size.numpy()
    o["scale/voxel_size"] = model.voxel_size.numpy()
    o["scale/sdf"] = model.sdf.grid.detach().numpy().copy()
    o["scale/off_color"] = model.off_color.grid.detach().numpy().copy()
    o["scale/nonempty_mask"] = model.nonempty_mask.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "host_dense_rows.npz"), **o)
    print("dense rows fixture:", {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in o.items() if "value" in k or "keep" in k},
          "kept", int(o["filter/keep/0"].sum()), int(o["filter/keep/1"].sum()), "world", o["scale/world_size"])


def lts_reference_loss(ns, results, rgbs, cfg):
    """Arithmetic of app/fine/lts.py:337-379 (no TV) with the reference's apply_gamma_curve."""
    tr = cfg.app.trainer
    loss = reference_loss(ns, results, rgbs, cfg)          # same first three terms, lts weights
    loss = loss + tr.weight_lts * F.mse_loss(results["lin/pbr/off"], results["lin/pbr/off_hat"])
    loss = loss + tr.weight_lts * F.mse_loss(results["lin/pbr/emo"], results["lin/pbr/emo_hat"])
    loss = loss + tr.weight_normal_smooth * F.l1_loss(results["etc/normal"], results["etc/normal_eps"])
    return loss


def gen_lts(ns, mask="full", sampling="random", neus_alpha="interp"):
    """ESRNeRF.forward_training (lts and pdra mode) on the small oblique slab, with every random draw
    of the reference recorded so that restatements can be fed the same numbers.  ``sampling="fib"``: the
    ``ray_sampling: fib`` variant (esrnerf.py:188-192; deterministic Fibonacci-spiral scattering directions)."""
    from esr_nerf_amd.config import lts_cfg
    cfg = lts_cfg("cpu", num_2ndrays=8, num_ltspts=12, ray_sampling=sampling, neus_alpha=neus_alpha)
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
    init_slab_model(model, sc)
    with torch.no_grad():
        model.brdf.grid.data.copy_(torch.randn(model.brdf.grid.shape, generator=torch.Generator().manual_seed(9)) * 0.1)
    model.train()
    if mask == "full":
        np.savez_compressed(os.path.join(OUT, "lts_g16_params.npz"),
                            **{k: v.detach().numpy() for k, v in model.state_dict().items()})
    else:
        with np.load(os.path.join(OUT, "lts_g16_params.npz")) as z:
            for k, v in model.state_dict().items():
                assert np.array_equal(z[k], v.detach().numpy()), k     # same parameters under every mask variant
    b = dict(sc.batch)
    b["uncert_masks"] = (torch.arange(sc.n_rays) % 3 == 0)
    for mode in (("lts", "pdra") if (sampling == "random" and neus_alpha == "interp") else ("lts",)):
        model.pdra_mode = mode == "pdra"
        model.zero_grad(set_to_none=True)
        rec = {"randn": [], "randn_like": []}
        r_randn, r_like, r_choice = torch.randn, torch.randn_like, np.random.choice

        def p_randn(*a, **k):
            t = r_randn(*a, **k)
            rec["randn"].append(t.clone())
            return t

        def p_like(x, **k):
            t = r_like(x, **k)
            rec["randn_like"].append(t.clone())
            return t

        def p_choice(*a, **k):
            v = r_choice(*a, **k)
            rec["idx"] = np.array(v)
            return v

        torch.manual_seed(5)
        np.random.seed(5)
        torch.randn, torch.randn_like, np.random.choice = p_randn, p_like, p_choice
        try:
            res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"],
                        uncert_masks=b["uncert_masks"], s_val=60.0, normal_eps=cfg.app.trainer.normal_eps,
                        emit_eps=cfg.app.trainer.emit_eps)
        finally:
            torch.randn, torch.randn_like, np.random.choice = r_randn, r_like, r_choice
        res_raw = {k: v.detach().clone() for k, v in res.items()}
        loss = lts_reference_loss(ns, dict(res), b["rgbs"], cfg)
        loss.backward()
        out = {"in/" + k: v.numpy() for k, v in b.items()}
        out["in/s_val"] = np.float32(60.0)
        out["draw/idx"] = rec["idx"].astype(np.int64)
        if sampling == "random":
            out["draw/dirs"] = rec["randn"][0].numpy()
        out["draw/noise_normal"] = rec["randn_like"][0].numpy()
        out["draw/noise_emit"] = rec["randn_like"][1].numpy()
        assert len(rec["randn"]) == (1 if sampling == "random" else 0) and len(rec["randn_like"]) == 2
        for k, v in res_raw.items():
            out["out/" + k] = v.numpy()
        out["loss"] = loss.detach().numpy()
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["grad/" + k] = p.grad.detach().numpy()
        tail = ("" if sampling == "random" else "_" + sampling) + ("" if neus_alpha == "interp" else "_gradalpha")
        np.savez_compressed(os.path.join(OUT, f"lts_g16_{mode}{_sfx(mask)}{tail}.npz"), **out)
        print("lts", mode, "loss", float(loss), "M3", res_raw["etc/normal"].shape[0],
              "grads", sum(1 for k in out if k.startswith("grad/")))


def gen_finetune(ns, mask="full"):
    """ESRNeRF.forward_finetune (re-lighting fine-tune, esrnerf.py:241-484) + the loss line of
    pdra.py:1090-1093 on the same small slab; parameters = lts_g16_params.npz with emo_color perturbed after
    train(finetune=True) froze its copy emit_color (so the two grids differ, as they do during fine-tuning)."""
    from esr_nerf_amd.config import lts_cfg
    cfg = lts_cfg("cpu", num_2ndrays=8, num_ltspts=12)
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
    init_slab_model(model, sc)
    with torch.no_grad():
        model.brdf.grid.data.copy_(torch.randn(model.brdf.grid.shape, generator=torch.Generator().manual_seed(9)) * 0.1)
    with np.load(os.path.join(OUT, "lts_g16_params.npz")) as z:
        for k, v in model.state_dict().items():
            assert np.array_equal(z[k], v.detach().numpy()), k        # same parameters as the lts fixtures
    for p_ in model.parameters():
        p_.requires_grad_(False)
    for p_ in list(model.emo_color.parameters()) + list(model.emo_rgbnet.parameters()):
        p_.requires_grad_(True)
    model.s_val = 60.0
    model.train(True, finetune=True)
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        model.emo_color.grid.add_(torch.randn(model.emo_color.grid.shape, generator=g) * 0.05)
    n = sc.n_rays
    b = dict(rays_o=sc.batch["rays_o"], rays_d=sc.batch["rays_d"], viewdirs=sc.batch["viewdirs"],
             em_modes=(torch.arange(n) % 5).long(), em_intensities=0.25 + 2.0 * torch.rand(n, generator=g),
             em_colors=torch.rand(n, 2, generator=g))
    rec = {}
    r_randn, r_choice = torch.randn, np.random.choice

    def p_randn(*a, **k):
        t = r_randn(*a, **k)
        rec.setdefault("randn", []).append(t.clone())
        return t

    def p_choice(*a, **k):
        v = r_choice(*a, **k)
        rec["idx"] = np.array(v)
        return v

    torch.manual_seed(6)
    np.random.seed(6)
    torch.randn, np.random.choice = p_randn, p_choice
    try:
        res = model(**b)
    finally:
        torch.randn, np.random.choice = r_randn, r_choice
    assert len(rec["randn"]) == 1
    w = 0.5                                                      # cfg/app/pdra.yaml:138 (eval.weight_lts)
    loss = w * F.mse_loss(res["lin/pbr/emo"], res["lin/pbr/emo_hat"])
    loss.backward()
    out = {"in/" + k: v.numpy() for k, v in b.items()}
    out["in/s_val"] = np.float32(60.0)
    out["in/weight_lts"] = np.float32(w)
    out["param/emo_color.grid"] = model.emo_color.grid.detach().numpy()
    out["param/emit_color.grid"] = model.emit_color.grid.detach().numpy()
    out["draw/idx"] = rec["idx"].astype(np.int64)
    out["draw/dirs"] = rec["randn"][0].numpy()
    for k, v in res.items():
        out["out/" + k] = v.detach().numpy()
    out["loss"] = loss.detach().numpy()
    for k, p_ in model.named_parameters():
        if p_.grad is not None:
            out["grad/" + k] = p_.grad.detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"lts_g16_finetune{_sfx(mask)}.npz"), **out)
    print("finetune loss", float(loss), "grads", sorted(k for k in out if k.startswith("grad/")))


def gen_lts_evals(ns, mask="full"):
    """ESRNeRF.eval_emit / eval_esp (PDRA regrouping queries, esrnerf.py:1299-1407) in eval mode on the oblique
    slab, parameters = lts_g16_params.npz, inputs = the rays of lts_g16_lts.npz."""
    from esr_nerf_amd.config import lts_cfg
    cfg = lts_cfg("cpu", num_2ndrays=8, num_ltspts=12)
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
    init_slab_model(model, sc)
    with torch.no_grad():
        model.brdf.grid.data.copy_(torch.randn(model.brdf.grid.shape, generator=torch.Generator().manual_seed(9)) * 0.1)
    with np.load(os.path.join(OUT, "lts_g16_params.npz")) as z:
        for k, v in model.state_dict().items():
            assert np.array_equal(z[k], v.detach().numpy()), k
    model.s_val = 60.0
    model.eval()
    b = sc.batch
    with torch.no_grad():
        e = model.eval_emit(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"])
        p_ = model.eval_esp(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"])
    np.savez_compressed(os.path.join(OUT, f"lts_g16_evals{_sfx(mask)}.npz"), **{"out/eval_emit": e.numpy(), "out/eval_esp": p_.numpy()})
    print("lts evals", float(e.abs().max()), float(p_.abs().max()))


def gen_lts_eval(ns, mask="full"):
    """ESRNeRF.forward_evaluate: em_modes 1 with render_pbr (per-sample light transport, scattering draws
    recorded per chunk) and em_modes 0 without; parameters = lts_g16_params.npz."""
    from esr_nerf_amd.config import lts_cfg
    cfg = lts_cfg("cpu", num_2ndrays=8, num_ltspts=12)
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
    init_slab_model(model, sc)
    with torch.no_grad():
        model.brdf.grid.data.copy_(torch.randn(model.brdf.grid.shape, generator=torch.Generator().manual_seed(9)) * 0.1)
    with np.load(os.path.join(OUT, "lts_g16_params.npz")) as z:
        for k, v in model.state_dict().items():
            assert np.array_equal(z[k], v.detach().numpy()), k
    model.s_val = 60.0
    model.eval()
    b = sc.batch
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(2)))
    rec = []
    r_randn = torch.randn

    def p_randn(*a, **k):
        t = r_randn(*a, **k)
        rec.append(t.clone())
        return t

    torch.manual_seed(4)
    torch.randn = p_randn
    try:
        with torch.no_grad():
            res1 = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=1, pos_rt=q,
                         render_pbr=True, chunk_sz=150)
    finally:
        torch.randn = r_randn
    with torch.no_grad():
        res0 = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=0, pos_rt=q,
                     render_pbr=False, chunk_sz=150)
    out = {"in/pos_rt": q.numpy(), "in/s_val": np.float32(60.0), "in/far": np.float32(sc.far), "in/chunk_sz": np.int64(150)}
    for i, t in enumerate(rec):
        out[f"draw/dirs{i}"] = t.numpy()
    for k, v in res1.items():
        out["out1/" + k] = v.numpy()
    for k, v in res0.items():
        out["out0/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"lts_g16_eval{_sfx(mask)}.npz"), **out)
    print("lts eval: chunks", len(rec), "keys", len(res1), len(res0))


def gen_coarse(ns, mask="full", neus_alpha="interp"):
    """VoxurfC.forward_training + the loss lines of coarse.py:341-352 on the small oblique slab (A17).
    ``neus_alpha="grad"``: the cfg variant of voxurfc.py:171-174 (section SDFs extrapolated with the sampled gradient)."""
    from esr_nerf_amd.config import coarse_cfg
    from esr_nerf_amd.synthetic import analytic_sdf
    sc = slab_scene("g16", s_val=8.0, oblique=True, mask=mask)
    cfg = coarse_cfg("cpu", num_voxels=sc.num_voxels, neus_alpha=neus_alpha)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.VoxurfC(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, 8.0)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        model.sdf.grid.copy_(analytic_sdf([int(v) for v in model.world_size], sc.xyz_min, sc.xyz_max))
        model.off_color.grid.copy_(torch.randn(model.off_color.grid.shape, generator=g) * 0.1)
        model.emo_color.grid.copy_(torch.randn(model.emo_color.grid.shape, generator=g) * 0.1)
    model.train()
    if mask == "full":
        np.savez_compressed(os.path.join(OUT, "coarse_g16_params.npz"),
                            **{k: v.detach().numpy() for k, v in model.state_dict().items()})
    else:
        with np.load(os.path.join(OUT, "coarse_g16_params.npz")) as z:
            for k, v in model.state_dict().items():
                assert np.array_equal(z[k], v.detach().numpy()), k
    b = sc.batch
    for s_val in (8.0, 40.0):
        model.zero_grad(set_to_none=True)
        res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=s_val)
        raw = {k: v.detach().clone() for k, v in res.items()}
        srgb = (res["srgb/rgb"] + res["etc/white_bg"] * 1.0).clamp(min=0.0, max=1.0)
        loss = F.mse_loss(srgb, b["rgbs"])
        pout = res["etc/alphainv_cum"][..., -1].clamp(1e-6, 1 - 1e-6)
        loss = loss + cfg.app.trainer.weight_entropy_last * -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
        loss.backward()
        out = {"in/" + k: v.numpy() for k, v in b.items()}
        out["in/s_val"] = np.float32(s_val)
        for k, v in raw.items():
            out["out/" + k] = v.numpy()
        out["loss"] = loss.detach().numpy()
        for k, p_ in model.named_parameters():
            if p_.grad is not None:
                out["grad/" + k] = p_.grad.detach().numpy()
        tail = "" if neus_alpha == "interp" else "_gradalpha"
        np.savez_compressed(os.path.join(OUT, f"coarse_g16_s{int(s_val)}{_sfx(mask)}{tail}.npz"), **out)
        print("coarse s_val", s_val, "loss", float(loss), "grads", sum(1 for k in out if k.startswith("grad/")))


def gen_coarse_eval(ns, mask="full"):
    """VoxurfC.forward_evaluate for em_modes 0 and 1, parameters = coarse_g16_params.npz."""
    from esr_nerf_amd.config import coarse_cfg
    from esr_nerf_amd.synthetic import analytic_sdf
    sc = slab_scene("g16", s_val=8.0, oblique=True, mask=mask)
    cfg = coarse_cfg("cpu", num_voxels=sc.num_voxels)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.VoxurfC(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, 8.0)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        model.sdf.grid.copy_(analytic_sdf([int(v) for v in model.world_size], sc.xyz_min, sc.xyz_max))
        model.off_color.grid.copy_(torch.randn(model.off_color.grid.shape, generator=g) * 0.1)
        model.emo_color.grid.copy_(torch.randn(model.emo_color.grid.shape, generator=g) * 0.1)
    with np.load(os.path.join(OUT, "coarse_g16_params.npz")) as z:
        for k, v in model.state_dict().items():
            assert np.array_equal(z[k], v.detach().numpy()), k
    model.s_val = 40.0
    model.eval()
    b = sc.batch
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(2)))
    out = {"in/pos_rt": q.numpy(), "in/s_val": np.float32(40.0), "in/far": np.float32(sc.far)}
    for em in (0, 1):
        with torch.no_grad():
            res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=em, pos_rt=q)
        for k, v in res.items():
            out[f"out{em}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"coarse_g16_eval{_sfx(mask)}.npz"), **out)
    print("coarse eval keys", sorted(k for k in out if k.startswith("out0/")))


def gen_eval(ns, mask="full"):
    """VoxurfF.forward_evaluate (image rendering, voxurff.py:280-461) for em_modes 0 and 1 on the oblique slab,
    parameters = fine_g16_params.npz."""
    cfg = fine_cfg("cpu")
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    torch.manual_seed(0)
    np.random.seed(0)
    model = ns.VoxurfF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
                       sc.mask_density, 60.0, sc.num_voxels)
    init_slab_model(model, sc)
    with np.load(os.path.join(OUT, "fine_g16_params.npz")) as z:
        for k, v in model.state_dict().items():
            assert np.array_equal(z[k], v.detach().numpy()), k
    model.eval()
    b = sc.batch
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(2)))
    out = {"in/" + k: b[k].numpy() for k in ("rays_o", "rays_d", "viewdirs")}
    out["in/pos_rt"] = q.numpy()
    out["in/s_val"] = np.float32(60.0)
    out["in/far"] = np.float32(sc.far)
    for em in (0, 1):
        with torch.no_grad():
            res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=em, pos_rt=q)
        for k, v in res.items():
            out[f"out{em}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"fine_g16_eval{_sfx(mask)}.npz"), **out)
    print("eval keys", sorted(k for k in out if k.startswith("out0/")))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "fib":
        gen_lts(ref_import.load(), "prune", "fib")
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "host":
        gen_host(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lts_eval":
        gen_lts_eval(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarse_eval":
        gen_coarse_eval(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lts_evals":
        gen_lts_evals(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        gen_eval(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarse":
        gen_coarse(ref_import.load())
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "finetune":
        gen_finetune(ref_import.load())
    else:
        main()
