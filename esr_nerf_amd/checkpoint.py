"""Checkpoint wire format of the reference trainers, so that stage-to-stage runs
(coarse -> fine -> lts -> pdra) and evaluation can exchange files with the reference in both directions.

Layout (app/fine/fine.py:466-490, the same in coarse.py / lts.py / pdra.py)::

    {"renderer": {"cfg", "near", "far", "xyz_min", "xyz_max", "mask_xyz_min", "mask_xyz_max",
                  "mask_alpha_init", "mask_density", "s_val", "num_voxels", "params": state_dict},
     "trainer":  {"global_step", "batch_st", "data_idxs", "optimizer": optimizer.state_dict()}}

``state_dict`` keys and LOGICAL shapes of the drop-in renderers equal the reference's (tests/test_host.py);
the colour grids are *stored* channels-last here, so a reference file is copied into that storage on load
(``load_state_dict`` keeps the destination's strides) and a file written here is made contiguous first --
either side reads ``[1, C, X, Y, Z]`` tensors.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F

RENDERER_KEYS = ("near", "far", "xyz_min", "xyz_max", "mask_xyz_min", "mask_xyz_max", "mask_alpha_init",
                 "mask_density", "s_val", "num_voxels")


def renderer_record(renderer) -> Dict[str, Any]:
    """The "renderer" entry (fine.py:468-481).  VoxurfC has no ``num_voxels`` constructor argument but the coarse
    trainer stores the attribute all the same."""
    rec = {"cfg": renderer.cfg}
    for k in RENDERER_KEYS:
        rec[k] = getattr(renderer, k)
    rec["params"] = {k: v.detach().contiguous() for k, v in renderer.state_dict().items()}
    return rec


def save_checkpoint(path: str, renderer, global_step: int, sampler=None, optimizer=None,
                    trainer_extra: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    trainer = {"global_step": global_step}
    if sampler is not None:
        if hasattr(sampler, "uncert_data_idxs"):             # pdra.py: the two ray groups
            trainer.update(uncert_batch_st=sampler.uncert_batch_st, cert_batch_st=sampler.cert_batch_st,
                           uncert_data_idxs=sampler.uncert_data_idxs, cert_data_idxs=sampler.cert_data_idxs)
        else:
            trainer.update(batch_st=sampler.batch_st, data_idxs=sampler.data_idxs)
    if optimizer is not None:
        trainer["optimizer"] = optimizer.state_dict()
    if trainer_extra:
        trainer.update(trainer_extra)
    ckpt = {"renderer": renderer_record(renderer), "trainer": trainer}
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path: str, device) -> Dict[str, Any]:
    # the files hold config objects and tensors (the reference's torch.load default): not weights-only
    return torch.load(path, map_location=device, weights_only=False)


def build_renderer(cls, cfg, rec: Dict[str, Any], device, num_voxels: Optional[int] = None, load_params: bool = True):
    """Instantiate ``cls`` (VoxurfC / VoxurfF / ESRNeRF drop-in) from a "renderer" record the way the trainers do
    (fine.py:231-253): constructor arguments from the record, then ``load_state_dict``."""
    args = [cfg] + [rec[k] for k in RENDERER_KEYS[:-1]]
    import inspect
    # (the fine renderers spell the parameter `num_voxles`, as the reference does, voxurff.py:42)
    if any(n.startswith("num_vox") for n in inspect.signature(cls.__init__).parameters):
        args.append(rec["num_voxels"] if num_voxels is None else num_voxels)
    renderer = cls(*args).to(device)
    if load_params:
        missing = renderer.load_state_dict(rec["params"], strict=False)
        # a stage adds modules the previous stage did not have (lts: brdf, brdfnet, emitnet, envmap); anything
        # else missing or unexpected is an error, as with the reference's strict load inside one stage
        bad = [k for k in missing.unexpected_keys]
        if bad:
            raise RuntimeError(f"unexpected keys in checkpoint: {bad[:5]}")
    return renderer


def fine_from_coarse(cls, cfg, coarse_rec: Dict[str, Any], device, num_voxels: int, sdf_reduce: float = 1.0,
                     pg_scale=(), scale_ratio: float = 1.0):
    """Start of the fine stage from a coarse checkpoint (fine.py:150-199): the fine renderer is built at the
    pre-scaling resolution, the coarse SDF is divided by ``sdf_reduce``, resampled to the fine grid (trilinear,
    align_corners) and smoothed with the 5^3 Gaussian (sigma 1); the non-empty mask follows."""
    from .modules import Gaussian3DConv
    nv = int(num_voxels / (scale_ratio ** len(pg_scale))) if len(pg_scale) else num_voxels
    renderer = build_renderer(cls, cfg, coarse_rec, device, num_voxels=nv, load_params=False)
    sdf = coarse_rec["params"]["sdf.grid"].to(device) / sdf_reduce
    if sdf.shape != renderer.sdf.grid.shape:
        sdf = F.interpolate(sdf, size=tuple(int(v) for v in renderer.world_size), mode="trilinear", align_corners=True)
    smooth = Gaussian3DConv(ksize=5, sigma=1).to(device)
    with torch.no_grad():
        renderer.sdf.grid.copy_(smooth(sdf))
    renderer.set_nonempty_mask()
    renderer.sdf_random_init = False
    return renderer
