"""Minimal stand-in for the Hydra/OmegaConf config objects the renderers read.

The drop-in classes only use attribute access (``cfg.app.model.stepsize`` ...), so
an OmegaConf ``DictConfig`` produced by the reference's ``run.py`` works unchanged;
this module provides the same tree without Hydra for tests and ``bench.py``.
Values are those of /root/reference/cfg/app/fine.yaml:13-30 and
cfg/__init__.yaml:20-27.
"""
from __future__ import annotations


class AttrDict(dict):
    """dict with attribute access, recursively."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                self[k] = AttrDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


FINE_MODEL = dict(
    mask_ks=3,
    maskcache_thres=0.001,
    fastcolor_thres=0.0001,
    stepsize=0.5,
    color_dim=6,
    rgbnet_width=192,
    rgbnet_depth=4,
    tonemap_width=192,
    tonemap_depth=2,
    posbase_pe=5,
    viewbase_pe=1,
    colorbase_pe=5,
    grad_feat=[0.5, 1.0, 1.5, 2.0],
    neus_alpha="interp",
)

FINE_TRAINER = dict(
    weight_entropy_last=0.001,
    weight_tv_density=0.01,
    weight_linear=0.1,
    tvs=dict(sdf=0.1, smooth_grad=0.05),
    s_start=20.0,
    s_inv_ratio=100.0,
)


# /root/reference/cfg/app/lts.yaml:12-41 (model) and :83-89 (loss weights)
LTS_MODEL = dict(FINE_MODEL, brdfnet_width=128, brdfnet_depth=4, env_sg=48, env_activation="softplus",
                 ray_sampling="random", num_2ndrays=256, num_ltspts=100, lts_near=1e-5)
LTS_TRAINER = dict(weight_entropy_last=0.001, weight_tv_density=0.01, weight_linear=10.0, weight_lts=0.01,
                   weight_normal_smooth=0.001, normal_eps=0.01, emit_eps=0.001, s_start=220.0)


# /root/reference/cfg/app/pdra.yaml:88-98
PDRA_TRAINER = dict(LTS_TRAINER, weight_emit_smooth=0.1, weight_lts_l=50.0, weight_lts_r=1.0, weight_emit_supp=0.1)


# /root/reference/cfg/app/coarse.yaml:12-32 (model), :75-77 (loss weights)
COARSE_MODEL = dict(mask_ks=3, maskcache_thres=0.001, fastcolor_thres=0.0001, stepsize=0.5, num_voxels=884736,
                    color_dim=12, rgbnet_width=128, rgbnet_depth=3, posbase_pe=5, viewbase_pe=1, smooth_ksize=5,
                    smooth_sigma=0.8, neus_alpha="interp")
COARSE_TRAINER = dict(weight_entropy_last=0.001, weight_tv_density=0.001, weight_tv_color=0.01,
                      tvs=dict(sdf=0.1, smooth_grad=0.05), s_start=5.0, s_inv_ratio=50.0)


def coarse_cfg(device: str = "cpu", **model_over) -> AttrDict:
    m = dict(COARSE_MODEL)
    m.update(model_over)
    return AttrDict(
        system=dict(device=device, debug=True, seed=0, tqdm_iters=10),
        app=dict(model=m, trainer=dict(COARSE_TRAINER)),
        data=dict(white_bg=True),
        global_step=0,
    )


def lts_cfg(device: str = "cpu", **model_over) -> AttrDict:
    m = dict(LTS_MODEL)
    m.update(model_over)
    return AttrDict(
        system=dict(device=device, debug=True, seed=0, tqdm_iters=10),
        app=dict(model=m, trainer=dict(PDRA_TRAINER)),      # superset of the lts keys
        data=dict(white_bg=True),
        global_step=0,
    )


def fine_cfg(device: str = "cpu", **model_over) -> AttrDict:
    return AttrDict(
        system=dict(device=device, debug=True, seed=0, tqdm_iters=10),
        app=dict(model=dict(FINE_MODEL, **model_over), trainer=dict(FINE_TRAINER)),
        data=dict(white_bg=True),
        global_step=0,
    )
