"""Teacher-student harness behind the bf16 PSNR bar (BASELINE.json: "PSNR within 0.1 dB of reference").

A fixed *teacher* parameter set renders the targets; a *student* with other parameters is trained on them with the
trainer-step objects of esr_nerf_amd.trainer and the fused Adam, once with f32 MLP operands (the reference's arithmetic)
and once with bf16 operands -- same seeds, same batches -- and both are scored by PSNR (utils2/metric.py:91-92:
-10 log10 MSE) on HELD-OUT rays of the teacher's image, rendered with the image-rendering entry points
(forward_evaluate).  Unlike a comparison of two renders of an untrained model against random targets (round 2), this
is sensitive: the score moves by tens of dB over training, and a precision problem in the bf16 forward, input
gradients or weight gradients shows up as a student that learns less.

Test infrastructure (tests/ and tools/ import it); nothing here is part of the product path.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

LRS_FINE = dict(off_color=0.1, off_rgbnet=0.003, emo_color=0.1, emo_rgbnet=0.003, sdf=0.005, tonemapper=0.003)
LRS_LTS = dict(LRS_FINE, brdf=0.1, brdfnet=0.003, emitnet=0.003, envmap=0.003)
TVS = dict(sdf=0.1, smooth_grad=0.05)


_TEACHER_CACHE: Dict[tuple, tuple] = {}


def paired_stats(d, conf: float = 0.95) -> Dict[str, float]:
    """mean, standard deviation and the ``conf`` confidence interval of the mean (Student t) of paired differences
    (keys keep the name ci95 whatever the level; ``conf`` is returned)."""
    d = np.asarray(list(d), dtype=np.float64)
    n = len(d)
    mean, sd = float(d.mean()), float(d.std(ddof=1)) if n > 1 else float("nan")
    # two-sided 97.5 % quantile of Student's t with n-1 degrees of freedom (scipy is importable in this image)
    from scipy import stats
    half = float(stats.t.ppf(0.5 + conf / 2.0, n - 1) * sd / math.sqrt(n)) if n > 1 else float("nan")
    return dict(n=n, mean=mean, sd=sd, conf=conf, ci95_half_width=half, ci95=(mean - half, mean + half))


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    return -10.0 * math.log10(max(float(((a.double() - b.double()) ** 2).mean()), 1e-20))


def smooth_field(shape, coarse, amp, gen):
    """[1,C,X,Y,Z] random field with structure a student can learn from a few thousand rays: white noise on a coarse
    lattice, trilinearly upsampled (a per-voxel white-noise grid has nothing held-out rays could be predicted from)."""
    c = shape[1]
    lat = torch.randn(1, c, *coarse, generator=gen) * amp
    return torch.nn.functional.interpolate(lat, size=tuple(shape[2:]), mode="trilinear", align_corners=True).contiguous()


def build_fine(scene, mlp_seed: int, grid_seed: int, dtype: str, smooth_amp: float = 0.0, lattice=(6, 6, 4)):
    from esr_nerf_amd.config import fine_cfg
    from esr_nerf_amd.synthetic import init_slab_model
    from esr_nerf_amd.voxurff import VoxurfF
    torch.manual_seed(mlp_seed)
    np.random.seed(mlp_seed)
    m = VoxurfF(fine_cfg("cuda:0"), scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min,
                scene.mask_xyz_max, scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
    init_slab_model(m, scene, seed=grid_seed)
    if smooth_amp > 0:
        g = torch.Generator().manual_seed(grid_seed + 1000)
        with torch.no_grad():
            for name in ("off_color", "emo_color"):
                grid = getattr(m, name).grid
                grid.copy_(smooth_field(grid.shape, tuple(lattice), smooth_amp, g).to(grid.device))
    m.mlp_dtype = dtype
    return m


@torch.no_grad()
def render_image(model, rays: Dict[str, torch.Tensor], s_val: float, chunk: int = 4096) -> torch.Tensor:
    """sRGB image of ``rays`` through forward_evaluate (voxurff.py:280-461), each ray under its own emissive mode,
    white background added as the trainer does (fine.py:357-358) -> [N,3] in [0,1]."""
    model.s_val = s_val
    model.eval()
    n = rays["rays_o"].shape[0]
    out = torch.empty(n, 3, device=rays["rays_o"].device)
    eye = torch.eye(3, device=out.device)
    for em in (0, 1):
        idx = torch.nonzero(rays["em_modes"] == em).flatten()
        for lo in range(0, idx.numel(), chunk):
            i = idx[lo:lo + chunk]
            r = model(rays_o=rays["rays_o"][i].contiguous(), rays_d=rays["rays_d"][i].contiguous(),
                      viewdirs=rays["viewdirs"][i].contiguous(), em_modes=em, pos_rt=eye)
            out[i] = (r["srgb/rgb"] + r["etc/white_bg"]).clamp(0, 1)
    model.train()
    return out


def cosine_schedule(steps: int):
    """The trainers' learning-rate schedule (fine.py:410-415 with cfg/app/fine.yaml:66-69: no warm-up, half cosine down
    to 0 at n_iters): without it the student never settles and the held-out score at a fixed step is optimisation
    noise (+-1 dB at lr 0.1 on the colour grids)."""
    from esr_nerf_amd.config import AttrDict
    from esr_nerf_amd.optimizer import CosineLR
    return CosineLR(AttrDict(app=dict(trainer=dict(n_iters=steps, warm_up_iters=0, warm_up_min_ratio=1.0,
                                                   const_warm_up=True, cos_min_ratio=0))))


def train_fine(student, train_rays: Dict[str, torch.Tensor], s_val: float, steps: int, batch: int, seed: int,
               eval_at: List[int], test_rays: Dict[str, torch.Tensor], test_img: torch.Tensor, lrs=None,
               weight_linear: float = 0.1):
    """The fine-stage trainer's loop shape (fine.py:346-415): batch -> FineStep -> every third step the TV lines ->
    fused Adam -> cosine learning-rate decay.  Returns {step: held-out PSNR} at the requested steps and the loss curve."""
    from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
    from esr_nerf_amd.trainer import FineStep
    student.train()
    step = FineStep(student, weight_linear=weight_linear)
    opt = create_optimizer_or_freeze_model(student, **(lrs or LRS_FINE))
    sched = cosine_schedule(steps)
    g = torch.Generator().manual_seed(seed)
    n = train_rays["rays_o"].shape[0]
    scores, losses = {}, []
    for it in range(steps + 1):
        if it in eval_at:
            scores[it] = psnr(render_image(student, test_rays, s_val), test_img)
        if it == steps:
            break
        idx = torch.randperm(n, generator=g)[:batch].to(train_rays["rays_o"].device)
        b = {k: v[idx].contiguous() for k, v in train_rays.items()}
        loss, grads = step.forward_loss_backward(b, s_val)
        if it % 3 == 0:
            step.add_regularisers(loss, grads, batch, 0.01, TVS, True)
        step.assign_grads(grads)
        opt.step()
        f = sched.decay_factor
        for pg in opt.param_groups:
            pg["lr"] *= f
        losses.append(float(loss))
    return scores, losses


def _mfma_f32_variant(fn):
    """dtype "f32mfma": the f32 engine with every MLP product on the f32 MFMA pipe (ESR_SPLIT_FWD=0 for the engines built
    inside the experiment) -- the arithmetic of rounds 1-3, against which the split-fp16 radiance kernels ("f32") are
    compared the same way bf16 is."""
    import functools
    import os

    @functools.wraps(fn)
    def run(dtype, *args, **kw):
        if dtype != "f32mfma":
            return fn(dtype, *args, **kw)
        keep = os.environ.get("ESR_SPLIT_FWD")
        os.environ["ESR_SPLIT_FWD"] = "0"
        try:
            return fn("f32", *args, **kw)
        finally:
            if keep is None:
                os.environ.pop("ESR_SPLIT_FWD", None)
            else:
                os.environ["ESR_SPLIT_FWD"] = keep
    return run


@_mfma_f32_variant
def fine_experiment(dtype: str, steps: int = 300, n_train: int = 12288, n_test: int = 4096, batch: int = 2048,
                    s_val: float = 40.0, seed: int = 0, eval_at=None, lrs=None, perturb=None,
                    weight_linear: float = 0.1, lattice=(6, 6, 4), jitter: float = 0.0):
    """Teacher (f32, smooth colour grids, MLP seed 100) -> image on n_train + n_test oblique rays of the `small` slab;
    student (MLP seed 200 + seed, N(0, 0.1) colour grids) trained with ``dtype`` MLP operands.  Returns held-out PSNR.
    ``jitter``: the student's initial MLP weights are multiplied by 1 + jitter * N(0, 1), once.  With jitter = 1e-3 (the
    rms size of one bf16 rounding) a seed whose f32 outcome moves by more than the bar cannot resolve a bf16-vs-f32
    question at that bar, whatever the operand type (tests/test_gpu_psnr.py uses it to MEASURE which seeds bifurcate)."""
    from esr_nerf_amd.synthetic import slab_scene
    key = ("fine", n_train, n_test, s_val, tuple(lattice))
    if key not in _TEACHER_CACHE:          # the teacher's image is the same for every seed and dtype: rendered once per process
        sc_ = slab_scene("small", s_val=s_val, oblique=True, n_rays=n_train + n_test, seed=31)
        rays_ = {k: v.cuda() for k, v in sc_.batch.items() if k != "rgbs"}
        teacher = build_fine(sc_, 100, 100, "f32", smooth_amp=0.6, lattice=lattice)
        _TEACHER_CACHE[key] = (sc_, rays_, render_image(teacher, rays_, s_val))
        del teacher
    sc, rays, img = _TEACHER_CACHE[key]
    train = {k: v[:n_train].contiguous() for k, v in rays.items()}
    train["rgbs"] = img[:n_train].contiguous()
    test = {k: v[n_train:].contiguous() for k, v in rays.items()}
    if perturb is None:
        student = build_fine(sc, 200 + seed, 200 + seed, dtype)
        if jitter:
            gj = torch.Generator().manual_seed(900 + seed)
            with torch.no_grad():
                for name, p_ in student.named_parameters():
                    if name.startswith(("off_rgbnet", "emo_rgbnet", "tonemapper")):
                        p_.mul_((1.0 + jitter * torch.randn(p_.shape, generator=gj)).to(p_.device))
    else:
        # the teacher's own parameters, perturbed: colour grids + N(0, perturb[0]), MLP weights x (1 + N(0, perturb[1]))
        student = build_fine(sc, 100, 100, dtype, smooth_amp=0.6)
        gp = torch.Generator().manual_seed(600 + seed)
        with torch.no_grad():
            for name, p_ in student.named_parameters():
                if name.startswith(("off_color", "emo_color")):
                    p_.add_((torch.randn(p_.shape, generator=gp) * perturb[0]).to(p_.device))
                elif name.startswith(("off_rgbnet", "emo_rgbnet", "tonemapper")):
                    p_.mul_((1.0 + perturb[1] * torch.randn(p_.shape, generator=gp)).to(p_.device))
    eval_at = eval_at if eval_at is not None else [0, steps]
    scores, losses = train_fine(student, train, s_val, steps, batch, 7 + seed, eval_at, test, img[n_train:], lrs=lrs,
                                weight_linear=weight_linear)
    return scores, losses, float(img.std())


# ---- LTS / PDRA renderer (BASELINE config C5) -----------------------------------------------------------------
def build_lts(scene, mlp_seed: int, grid_seed: int, dtype: str, smooth_amp: float = 0.0, num_2ndrays: int = 16,
              num_ltspts: int = 24):
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.esrnerf import ESRNeRF
    from esr_nerf_amd.synthetic import init_slab_model
    torch.manual_seed(mlp_seed)
    np.random.seed(mlp_seed)
    cfg = lts_cfg("cuda:0", num_2ndrays=num_2ndrays, num_ltspts=num_ltspts)
    m = ESRNeRF(cfg, scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min, scene.mask_xyz_max,
                scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
    init_slab_model(m, scene, seed=grid_seed)
    g = torch.Generator().manual_seed(grid_seed + 1000)
    with torch.no_grad():
        if smooth_amp > 0:
            for name in ("off_color", "emo_color", "brdf"):
                grid = getattr(m, name).grid
                grid.copy_(smooth_field(grid.shape, (6, 6, 4), smooth_amp, g).to(grid.device))
        else:
            m.brdf.grid.copy_((torch.randn(m.brdf.grid.shape, generator=g) * 0.1).to(m.brdf.grid.device))
    m.mlp_dtype = dtype
    return m, cfg


@_mfma_f32_variant
def pdra_experiment(dtype: str, steps: int = 200, n_train: int = 6144, n_test: int = 2048, batch: int = 1024,
                    s_val: float = 60.0, seed: int = 0, eval_at=None, num_2ndrays: int = 256, num_ltspts: int = 100):
    """C5's first half: a student ``ESRNeRF`` trained with ``LtsStep(stage="pdra")`` (image loss + light-transport,
    emission-suppression and smoothness terms, pdra.py:374-475) + fused Adam + cosine decay on a teacher's image;
    held-out PSNR of its ``forward_evaluate`` image (esrnerf.py:1003-1297).

    This stage's training is CHAOTIC at the test's scale (measured on MI355X, 200 steps, 3 seeds): two f32 runs of the
    same seeds -- differing only in float-atomic ordering -- end 0.1-0.8 dB apart (the Monte-Carlo light-transport terms
    start 100x above the image term, and the step's draws -- np.random.choice(M3, P), randn(M3, 3) -- change with the
    survivor count, so one sample more or less re-seeds every later step); tying the draws to fixed uniforms or starting
    from a perturbed teacher did not change that.  A single run therefore cannot resolve 0.1 dB here; the test that
    uses this function checks that bf16 lands inside the f32 runs' band, and the 0.1 dB bar itself is asserted where
    training is reproducible (fine stage: f32 reruns within 0.05 dB; fine-tune half: identical).
    (The trainer's every-third-step ``do_tv`` lines, pdra.py:459-476, are not part of this loop -- in either arm: the committed
    statistics were drawn without them, and they act on the SDF grid alone, in fp32, identically for both operand types.)"""
    from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import LtsStep
    key = ("pdra", n_train, n_test, s_val)
    if key not in _TEACHER_CACHE:          # the teacher's image is the same for every seed and dtype: rendered once per process
        sc_ = slab_scene("small", s_val=s_val, oblique=True, n_rays=n_train + n_test, seed=37)
        rays_ = {k: v.cuda() for k, v in sc_.batch.items() if k != "rgbs"}
        teacher, _ = build_lts(sc_, 100, 100, "f32", smooth_amp=0.6)
        _TEACHER_CACHE[key] = (sc_, rays_, render_image(teacher, rays_, s_val))
        del teacher
    sc, rays, img = _TEACHER_CACHE[key]
    train = {k: v[:n_train].contiguous() for k, v in rays.items()}
    train["rgbs"] = img[:n_train].contiguous()
    test = {k: v[n_train:].contiguous() for k, v in rays.items()}
    # the stage's own estimator sizes (cfg/app/lts.yaml: 100 surface points x 256 secondary rays)
    student, cfg = build_lts(sc, 200 + seed, 200 + seed, dtype, num_2ndrays=num_2ndrays, num_ltspts=num_ltspts)
    student.train()
    student.pdra_mode = True
    opt = create_optimizer_or_freeze_model(student, **LRS_LTS)
    step = LtsStep(student, cfg.app.trainer, stage="pdra")
    sched = cosine_schedule(steps)
    g = torch.Generator().manual_seed(11 + seed)
    torch.manual_seed(300 + seed)
    np.random.seed(300 + seed)
    eval_at = eval_at if eval_at is not None else [0, steps]
    scores, losses = {}, []
    for it in range(steps + 1):
        if it in eval_at:
            st = torch.get_rng_state(), torch.cuda.get_rng_state(), np.random.get_state()
            scores[it] = psnr(render_image(student, test, s_val), img[n_train:])
            student.pdra_mode = True
            torch.set_rng_state(st[0]); torch.cuda.set_rng_state(st[1]); np.random.set_state(st[2])
        if it == steps:
            break
        idx = torch.randperm(n_train, generator=g)[:batch].cuda()
        b = {k: v[idx].contiguous() for k, v in train.items()}
        b["uncert_masks"] = (torch.arange(batch, device="cuda") % 3 != 0)
        loss, grads, _ = step.forward_loss_backward(b, s_val)
        step.assign_grads(grads)
        opt.step()
        f = sched.decay_factor
        for pg in opt.param_groups:
            pg["lr"] *= f
        losses.append(float(loss))
    return scores, losses, float(img.std())


@_mfma_f32_variant
def finetune_experiment(dtype: str, steps: int = 80, n_rays: int = 2048, n_test: int = 2048, s_val: float = 60.0,
                        seed: int = 0, eval_at=None):
    """C5's second half (pdra.py:1047-1109): from a fixed parameter set only ``emo_color`` / ``emo_rgbnet`` train, towards
    the edited emission + its light transport: loss 0.5 * MSE(lin/pbr/emo, lin/pbr/emo_hat).  Score, as the reference
    reports it (loss2psnr, utils2/metric.py:91-92): PSNR of that MSE on HELD-OUT rays with fixed draws, evaluated with
    the f32 engine for both students (one scorer), plus the ``forward_evaluate`` image of each for the direct
    comparison.  Returns ({step: psnr}, losses, image [n_test,3])."""
    from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("small", s_val=s_val, oblique=True, n_rays=n_rays + n_test, seed=41)
    rays = {k: v.cuda() for k, v in sc.batch.items() if k != "rgbs"}
    m, cfg = build_lts(sc, 100, 100, dtype, smooth_amp=0.6)
    for p in m.parameters():
        p.requires_grad_(False)
    for p in list(m.emo_color.parameters()) + list(m.emo_rgbnet.parameters()):
        p.requires_grad_(True)
    m.s_val = s_val
    m.train(True, finetune=True)
    ge = torch.Generator().manual_seed(21)
    n_all = n_rays + n_test
    edit = dict(em_modes=(torch.arange(n_all) % 5).long().cuda(),
                em_intensities=(0.25 + 2.0 * torch.rand(n_all, generator=ge)).cuda(),
                em_colors=torch.rand(n_all, 2, generator=ge).cuda())
    pick = lambda lo, hi: dict(rays_o=rays["rays_o"][lo:hi].contiguous(), rays_d=rays["rays_d"][lo:hi].contiguous(),
                               viewdirs=rays["viewdirs"][lo:hi].contiguous(),
                               **{k: v[lo:hi].contiguous() for k, v in edit.items()})
    train, test = pick(0, n_rays), pick(n_rays, n_all)
    opt = create_optimizer_or_freeze_model(m, emo_color=0.1, emo_rgbnet=0.003)
    sched = cosine_schedule(steps)

    def held_out():
        st = torch.get_rng_state(), torch.cuda.get_rng_state(), np.random.get_state()
        torch.manual_seed(999); np.random.seed(999)
        keep = m.mlp_dtype
        if keep != "f32":                    # one scorer for both students
            m.mlp_dtype = "f32"; m._engine = None
        with torch.no_grad():
            r = m(**test)
            v = -10.0 * math.log10(max(float(((r["lin/pbr/emo"] - r["lin/pbr/emo_hat"]) ** 2).mean()), 1e-20))
        if keep != "f32":
            m.mlp_dtype = keep; m._engine = None
        torch.set_rng_state(st[0]); torch.cuda.set_rng_state(st[1]); np.random.set_state(st[2])
        return v

    torch.manual_seed(400 + seed)
    np.random.seed(400 + seed)
    eval_at = eval_at if eval_at is not None else [0, steps]
    scores, losses = {}, []
    for it in range(steps + 1):
        if it in eval_at:
            scores[it] = held_out()
        if it == steps:
            break
        m.zero_grad(set_to_none=True)
        r = m(**train)
        loss = 0.5 * torch.nn.functional.mse_loss(r["lin/pbr/emo"], r["lin/pbr/emo_hat"])
        loss.backward()
        opt.step()
        f = sched.decay_factor
        for pg in opt.param_groups:
            pg["lr"] *= f
        losses.append(float(loss))
    m.train(False)
    with torch.no_grad():
        img = render_image(m, {k: test[k] for k in ("rays_o", "rays_d", "viewdirs")} | {"em_modes": (test["em_modes"] > 0).long()}, s_val)
    return scores, losses, img
