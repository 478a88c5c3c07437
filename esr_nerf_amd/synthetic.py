"""Synthetic "slab" scenes of SURVEY.md section 8(d) / BASELINE.md section 2.

The reference's sampler draws ``ceil((t_max - t_min) * |d| / (0.5 * voxel))`` steps
per ray (app/utils/base/cuda/render_utils_kernel.cu:53).  In a box
``(-1,-1,-z) .. (1,1,z)`` with ``voxel = 2/256`` an axis-parallel ray therefore
gets exactly ``256 * z * 2`` samples, which pins BASELINE.json's
"4096 rays x 128 samples" (z = 0.25) and "x 192 samples" (z = 0.375) on the
reference's own arithmetic.  Everything is seeded and generated on CPU, then
moved to the requested device, so the CPU oracle and the HIP path see identical
bits.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import torch

# name -> (n_rays, z half-extent, voxels along x)   [z-extent in voxels = res * z]
CONFIGS = {
    "C1": dict(n_rays=512, z=1.0 / 3.0, res=96),      # coarse plumbing size, ~64 samples
    "C2": dict(n_rays=4096, z=0.25, res=256),         # fine, 128 samples (north star)
    "C3": dict(n_rays=4096, z=0.375, res=256),        # fine, 192 samples
    "C4": dict(n_rays=8192, z=0.25, res=256),         # lts / pdra stages: 8192 primary rays (+100x256 secondary)
    "tiny": dict(n_rays=64, z=0.25, res=64),          # 32 samples, golden-vector size
    "small": dict(n_rays=512, z=0.25, res=128),       # 64 samples
    "g16": dict(n_rays=48, z=0.25, res=32),           # 16 samples, committed golden fixtures
    # the fine stage's STARTING resolution class (cfg/app/fine.yaml:41-43: 160^3 until step 15000, then 256^3): the slab
    # at 160 voxels along x -- world [160,160,40]; tests/test_gpu_scale_event.py scales it to [256,256,64] mid-run
    "g160": dict(n_rays=1024, z=0.25, res=160),
    # production-size grid (cfg/app/fine.yaml:41-43: the fine stage ends at 256^3): the cube (-1,-1,-1)..(1,1,1) at 256
    # voxels per axis; the mask cache's box is the slab |z| < 0.25 (the occupied part of a real scene is a fraction of
    # its box), so an axis-parallel ray walks 512 steps through the box and keeps C2's 128 samples
    "C2g256": dict(n_rays=4096, z=1.0, res=256, mask_z=0.25),
}


@dataclass
class SlabScene:
    name: str
    n_rays: int
    xyz_min: torch.Tensor
    xyz_max: torch.Tensor
    num_voxels: int
    near: float
    far: float
    mask_density: torch.Tensor        # [1,1,32,32,32]
    mask_alpha_init: float
    s_val: float
    mask_xyz_min: torch.Tensor = None  # box of the mask cache (fine.py:149-184 takes it from the coarse stage's alpha mask)
    mask_xyz_max: torch.Tensor = None
    batch: Dict[str, torch.Tensor] = field(default_factory=dict)   # rays_o, rays_d, viewdirs, em_modes, rgbs
    sdf_fn: object = None


def prune_mask(xyz_min: torch.Tensor, xyz_max: torch.Tensor, seed: int = 0):
    """A mask cache that prunes: box strictly inside the scene box (samples between the two
    boxes read zero padding), a smooth random density field that crosses the occupancy threshold
    (alpha >= 1e-3  <=>  density >= 6.908 at mask_alpha_init = 1e-6) at non-grid positions, an EMPTY z-slab just in
    front of the surface (a ray's survivors are non-contiguous steps: the NeuS neighbour rule of functions.py:80-98
    pairs the samples either side of the gap), and a sub-threshold column (whole rays without a survivor).
    Returns (mask_density [1,1,32,32,32], mask_xyz_min, mask_xyz_max); the model constructor max-pools the density
    with mask_ks = 3 (module.py:95-100), so every feature is at least 5 cells wide."""
    g = torch.Generator().manual_seed(seed + 77)
    ext = xyz_max - xyz_min
    lo = xyz_min + ext * torch.tensor([0.06, 0.03, 0.08])
    hi = xyz_max - ext * torch.tensor([0.04, 0.07, 0.05])
    coarse = 2.0 + 12.0 * torch.rand(1, 1, 5, 5, 5, generator=g)
    dens = torch.nn.functional.interpolate(coarse, size=(32, 32, 32), mode="trilinear", align_corners=True)
    ax = [torch.linspace(float(lo[a]), float(hi[a]), 32) for a in range(3)]
    gx, gy, gz = torch.meshgrid(*ax, indexing="ij")
    zh = float(xyz_max[2])
    dens[0, 0][(gz > 0.08 * zh) & (gz < 0.48 * zh)] = 0.0                     # empty slab in front of the surface
    dens[0, 0][((gx - 0.3) ** 2 + (gy + 0.2) ** 2) < 0.25 ** 2] = 5.0         # occupied-looking but below threshold
    return dens.contiguous(), lo, hi


def slab_scene(name: str = "C2", s_val: float = 20.0, seed: int = 0, n_rays: int | None = None,
               oblique: bool = False, mask: str = "full") -> SlabScene:
    """Build inputs for one of the BASELINE configs.

    ``mask="prune"`` replaces the all-occupied mask cache by ``prune_mask`` (non-uniform density, mask box strictly
    inside the scene box).

    ``oblique=True`` tilts and jitters the ray directions (un-normalised ``rays_d``)
    so rays get ragged step counts and clip the box faces: the edge-case variant
    used by the parity tests; the bench uses the axis-parallel default.
    """
    c = CONFIGS[name]
    n = int(n_rays if n_rays is not None else c["n_rays"])
    z, res = float(c["z"]), int(c["res"])
    g = torch.Generator().manual_seed(seed)
    xyz_min = torch.tensor([-1.0, -1.0, -z])
    xyz_max = torch.tensor([1.0, 1.0, z])
    num_voxels = res * res * int(round(res * z))
    xy = (torch.rand(n, 2, generator=g) * 2 - 1) * 0.9
    rays_o = torch.cat([xy, torch.full((n, 1), 2.0)], -1)
    rays_d = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
    if oblique:
        tilt = (torch.rand(n, 2, generator=g) * 2 - 1) * 0.35
        rays_d = torch.cat([tilt, -torch.ones(n, 1)], -1)
        rays_d = rays_d * (0.5 + torch.rand(n, 1, generator=g))        # un-normalised on purpose
        rays_d[::17, 0] = 0.0                                           # exercise the d==0 branch
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    em_modes = (torch.arange(n) % 2).long()
    rgbs = torch.rand(n, 3, generator=g)
    rgbs[::13] = 1.0                                                    # exercise the rgbs>=1 branch of the loss
    if mask == "prune":
        mask_density, mlo, mhi = prune_mask(xyz_min, xyz_max, seed)
    else:
        assert mask == "full", mask
        mask_density, mlo, mhi = torch.full((1, 1, 32, 32, 32), 30.0), xyz_min.clone(), xyz_max.clone()
        if "mask_z" in c:                                               # mask box = a slab inside the scene box
            mlo[2], mhi[2] = -float(c["mask_z"]), float(c["mask_z"])
    return SlabScene(
        name=name, n_rays=n, xyz_min=xyz_min, xyz_max=xyz_max, num_voxels=num_voxels,
        near=0.05, far=6.0, mask_density=mask_density, mask_xyz_min=mlo, mask_xyz_max=mhi,
        mask_alpha_init=1e-6, s_val=float(s_val),
        batch=dict(rays_o=rays_o.contiguous(), rays_d=rays_d.contiguous(),
                   viewdirs=viewdirs.contiguous(), em_modes=em_modes, rgbs=rgbs),
    )


def analytic_sdf(world_size, xyz_min, xyz_max) -> torch.Tensor:
    """SDF = z + 0.05 sin(3x) cos(3y) sampled on the grid nodes -> [1,1,X,Y,Z]."""
    X, Y, Z = [int(v) for v in world_size]
    xs = torch.linspace(float(xyz_min[0]), float(xyz_max[0]), X)
    ys = torch.linspace(float(xyz_min[1]), float(xyz_max[1]), Y)
    zs = torch.linspace(float(xyz_min[2]), float(xyz_max[2]), Z)
    gx, gy, gz = torch.meshgrid(xs, ys, zs, indexing="ij")
    return (gz + 0.05 * torch.sin(3 * gx) * torch.cos(3 * gy))[None, None].contiguous()


@torch.no_grad()
def init_slab_model(model, scene: SlabScene, seed: int = 0):
    """Overwrite grids of a VoxurfF-compatible model (reference or ours) with the
    slab content: analytic SDF, N(0, 0.1) colour grids.  MLPs keep their default
    nn.Linear init (drawn by the caller under ``torch.manual_seed``)."""
    g = torch.Generator().manual_seed(seed + 1)
    ws = [int(v) for v in model.world_size]
    dev = model.sdf.grid.device
    model.sdf.grid.data.copy_(analytic_sdf(ws, scene.xyz_min, scene.xyz_max).to(dev))
    for name in ("off_color", "emo_color"):
        grid = getattr(model, name).grid
        vals = torch.randn(tuple(grid.shape), generator=g) * 0.1
        grid.data.copy_(vals.to(dev))
    model.sdf_random_init = False
    return model
