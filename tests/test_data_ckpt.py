"""Ray samplers and checkpoint wire format: the index-based, rank-aware samplers return
the reference's batches (checked against the imported reference classes when /root/reference is present, and
through self-consistency otherwise); checkpoints round-trip in the reference layout."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from esr_nerf_amd.config import AttrDict, fine_cfg
from esr_nerf_amd.data import BatchSampler, RayGroupManager

sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

KEYS = ["rays_o", "rgbs", "em_modes"]


def _cfg():
    return AttrDict(system=dict(device="cpu", data_preload="gpu"))      # the device-resident branch, on the CPU "device"


def _data(n=103, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"rays_o": torch.randn(n, 3, generator=g), "rgbs": torch.rand(n, 3, generator=g),
            "em_modes": torch.randint(0, 2, (n,), generator=g)}


def _ref_utils():
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.load()
    import importlib
    return importlib.import_module("utils2.utils")


def test_batch_sampler_matches_reference_batches():
    ru = _ref_utils()
    d = _data()
    mask = torch.arange(103) % 5 != 0

    def run(cls, data):
        torch.manual_seed(5)                               # both see the same global generator sequence
        s = cls(_cfg(), data, KEYS, 16)
        s.filter(mask)
        s.shuffle()
        return s, [s.sample() for _ in range(20)], s.batch_st      # crosses three epoch boundaries

    ref, a, st_a = run(ru.BatchSampler, {k: v.clone() for k, v in d.items()})
    ours, b, st_b = run(BatchSampler, d)
    for x, y in zip(a, b):
        for k in KEYS:
            assert torch.equal(x[k], y[k]), k
    assert st_a == st_b
    assert torch.equal(ref.data_idxs, ours.data_idxs) and ref.data_num == ours.data_num
    assert torch.equal(ref.data["rgbs"], ours.current("rgbs"))
    # resume: a sampler rebuilt from (batch_st, data_idxs) continues the same sequence (fine.py:220-227)
    again = BatchSampler(_cfg(), d, KEYS, 16, ours.batch_st, ours.data_idxs.clone())
    torch.manual_seed(9)
    x = ours.sample()
    torch.manual_seed(9)
    y = again.sample()
    assert all(torch.equal(x[k], y[k]) for k in KEYS)


def test_ray_group_manager_matches_reference_batches():
    ru = _ref_utils()
    d = _data(211, seed=1)

    def run(cls, data):
        torch.manual_seed(3)
        s = cls(_cfg(), data, KEYS, 24, 8)
        out = []
        for it in range(30):
            if it in (4, 17):                               # regrouping: some uncertain rays become certain
                m = torch.rand(s.uncert_data_num, generator=torch.Generator().manual_seed(it)) < 0.7
                s.filter(m)
                s.shuffle()
            out.append(s.sample())
        return s, out

    ref, a = run(ru.RayGroupManager, {k: v.clone() for k, v in d.items()})
    ours, b = run(RayGroupManager, d)
    for it, (x, y) in enumerate(zip(a, b)):
        assert set(x) == set(y)
        for k in x:
            assert torch.equal(x[k], y[k]), (it, k)        # includes the all-False mask while the certain group is empty
    assert torch.equal(ref.uncert_data_idxs, ours.uncert_data_idxs)
    assert torch.equal(ref.cert_data_idxs, ours.cert_data_idxs)


@pytest.mark.parametrize("world", [2, 3])
def test_rank_shares_tile_the_global_batch(world):
    """Each rank is its own process with its own (identically seeded) generator: run every sampler's whole
    sequence under the same seed, then compare."""
    d = _data(97, seed=2)

    def run_bs(rank, w):
        torch.manual_seed(1)
        s = BatchSampler(_cfg(), d, KEYS, 24, rank=rank, world=w)
        return [s.sample() for _ in range(9)]

    whole = run_bs(0, 1)
    parts = [run_bs(r, world) for r in range(world)]
    for i, g in enumerate(whole):
        for k in KEYS:
            assert torch.equal(torch.cat([p[i][k] for p in parts]), g[k])

    keep = torch.arange(97) % 3 != 0

    def run_gm(rank, w):
        torch.manual_seed(2)
        s = RayGroupManager(_cfg(), d, KEYS, 12, 6, rank=rank, world=w)
        s.filter(keep)
        return [s.sample() for _ in range(6)]

    whole = run_gm(0, 1)
    parts = [run_gm(r, world) for r in range(world)]
    for i, g in enumerate(whole):
        # every rank carries the global uncertain : certain mix; the union is the global batch
        unc = torch.cat([p[i]["rgbs"][p[i]["uncert_masks"]] for p in parts])
        cer = torch.cat([p[i]["rgbs"][~p[i]["uncert_masks"]] for p in parts])
        assert torch.equal(unc, g["rgbs"][g["uncert_masks"]]) and torch.equal(cer, g["rgbs"][~g["uncert_masks"]])


def _cpu_model():
    from esr_nerf_amd.synthetic import init_slab_model, slab_scene
    from esr_nerf_amd.voxurff import VoxurfF
    sc = slab_scene("g16", s_val=20.0)
    torch.manual_seed(0)
    np.random.seed(0)
    m = VoxurfF(fine_cfg("cpu"), sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
                sc.mask_density, sc.s_val, sc.num_voxels)
    return init_slab_model(m, sc), sc


def test_checkpoint_round_trip_in_reference_layout(tmp_path):
    from esr_nerf_amd import checkpoint as ck
    from esr_nerf_amd.voxurff import VoxurfF
    m, sc = _cpu_model()
    with torch.no_grad():
        m.emo_color.grid.normal_(0, 0.3)
    sampler = BatchSampler(_cfg(), _data(), KEYS, 16)
    sampler.shuffle()
    p = str(tmp_path / "last.ckpt")
    ck.save_checkpoint(p, m, 41, sampler=sampler)
    z = ck.load_checkpoint(p, "cpu")
    assert set(z) == {"renderer", "trainer"}
    assert set(z["renderer"]) == {"cfg", "near", "far", "xyz_min", "xyz_max", "mask_xyz_min", "mask_xyz_max",
                                  "mask_alpha_init", "mask_density", "s_val", "num_voxels", "params"}
    assert z["trainer"]["global_step"] == 41 and torch.equal(z["trainer"]["data_idxs"], sampler.data_idxs)
    # what a reference reader sees: plain contiguous [1, C, X, Y, Z] tensors under the reference's key names
    g = z["renderer"]["params"]["emo_color.grid"]
    assert g.shape == m.emo_color.grid.shape and g.is_contiguous()
    m2 = ck.build_renderer(VoxurfF, fine_cfg("cpu"), z["renderer"], "cpu")
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    assert m2.emo_color.grid.is_contiguous(memory_format=torch.channels_last_3d)     # storage of THIS build
    assert m2.world_size.tolist() == m.world_size.tolist()


def test_fine_stage_starts_from_a_coarse_record():
    """fine.py:150-199: SDF / sdf_reduce, trilinear resample to the fine grid, 5^3 Gaussian; pre-scaling resolution."""
    from esr_nerf_amd import checkpoint as ck
    from esr_nerf_amd.modules import Gaussian3DConv
    from esr_nerf_amd.voxurff import VoxurfF
    import torch.nn.functional as F
    m, sc = _cpu_model()
    rec = ck.renderer_record(m)
    coarse_sdf = torch.randn(1, 1, 12, 12, 4)
    rec["params"] = {"sdf.grid": coarse_sdf}
    fine = ck.fine_from_coarse(VoxurfF, fine_cfg("cpu"), rec, "cpu", num_voxels=sc.num_voxels * 8, sdf_reduce=2.0,
                               pg_scale=[100], scale_ratio=8.0)
    assert fine.num_voxels == sc.num_voxels and fine.sdf_random_init is False
    want = Gaussian3DConv(5, 1)(F.interpolate(coarse_sdf / 2.0, size=tuple(fine.sdf.grid.shape[2:]), mode="trilinear",
                                              align_corners=True))
    want[~fine.nonempty_mask] = 1
    assert torch.allclose(fine.sdf.grid, want)
