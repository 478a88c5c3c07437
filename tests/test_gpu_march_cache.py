"""The three march passes with a shared cache (esr_fine_march_*_cached: one walk per step) against the uncached entry
points on the same rays: records bit-identical, SDF gradient equal up to the summation order of float atomics, and the
per-record value-tap array (dsdf_rec) + direct scatter equal to the uncached backward's scatter."""
import ctypes as C

import pytest
import torch

from test_gpu_fine_path import build_gpu_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mask,s_val,oblique", [("full", 20.0, False), ("prune", 60.0, True), ("prune", 220.0, True)])
def test_cached_march_equals_uncached(mask, s_val, oblique):
    from esr_nerf_amd import _lib
    from esr_nerf_amd.synthetic import slab_scene
    L = _lib.lib()
    sc = slab_scene("small", s_val=s_val, oblique=oblique, n_rays=300, seed=4, mask=mask)
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    m.s_val = s_val
    scene = m.scene_struct()
    sp = C.byref(scene)
    dev = "cuda:0"
    rays_o, rays_d = sc.batch["rays_o"].to(dev).contiguous(), sc.batch["rays_d"].to(dev).contiguous()
    em = sc.batch["em_modes"].to(dev).contiguous()
    n = rays_o.shape[0]
    mask_d = m.mask_cache.density.view(*m.mask_cache.density.shape[2:]).contiguous()
    sdf = m.sdf.device_view()
    s = _lib.stream_ptr(dev)
    i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)
    out = {}
    for cached in (False, True):
        cnt3, off3, stats, last = i32(n), i32(n), i32(3 * n), torch.empty(n, device=dev)
        plan = torch.zeros(8, dtype=torch.int32, device=dev)
        cache = torch.empty(int(L.esr_fine_march_cache_floats(sp, n)), device=dev) if cached else None
        if cached:
            _lib.check(L.esr_fine_march_count_cached(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(mask_d), _lib.ptr(sdf), n,
                                                     _lib.ptr(cnt3), _lib.ptr(last), _lib.ptr(stats), _lib.ptr(plan),
                                                     _lib.ptr(cache), s), "count_cached")
        else:
            _lib.check(L.esr_fine_march_count(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(mask_d), _lib.ptr(sdf), n,
                                              _lib.ptr(cnt3), _lib.ptr(last), _lib.ptr(stats), _lib.ptr(plan), s), "count")
        _lib.check(L.esr_fine_plan(_lib.ptr(cnt3), _lib.ptr(em), _lib.ptr(stats), n, _lib.ptr(off3), _lib.ptr(plan), s), "plan")
        hdr = plan.tolist()
        tiles = hdr[3]
        assert tiles > 0 and hdr[7] == 0
        rec_ray = torch.full((tiles * 32,), -1, dtype=torch.int32, device=dev)
        rec_step, rec_w, rec_sdf = i32(tiles * 32).zero_(), torch.zeros(tiles * 32, device=dev), torch.zeros(tiles * 32, device=dev)
        if cached:
            _lib.check(L.esr_fine_march_fill_cached(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), n, _lib.ptr(off3), _lib.ptr(stats),
                                                    _lib.ptr(cache), _lib.ptr(rec_ray), _lib.ptr(rec_step), _lib.ptr(rec_w),
                                                    _lib.ptr(rec_sdf), s), "fill_cached")
        else:
            _lib.check(L.esr_fine_march_fill(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(mask_d), _lib.ptr(sdf), n,
                                             _lib.ptr(off3), _lib.ptr(rec_ray), _lib.ptr(rec_step), _lib.ptr(rec_w),
                                             _lib.ptr(rec_sdf), s), "fill")
        g = torch.Generator(device="cpu").manual_seed(7)
        dweight = torch.randn(tiles * 32, generator=g).to(dev)
        dlast = torch.randn(n, generator=g).to(dev)
        grad = torch.zeros_like(sdf)
        dsdf = torch.zeros(tiles * 32, device=dev)
        if cached:
            _lib.check(L.esr_fine_march_bwd_cached(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), n, _lib.ptr(off3), _lib.ptr(stats),
                                                   _lib.ptr(last), _lib.ptr(cache), _lib.ptr(dweight), _lib.ptr(dlast),
                                                   _lib.ptr(grad), _lib.ptr(dsdf), 0, s), "bwd_cached")
        else:
            _lib.check(L.esr_fine_march_bwd_rec(sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(mask_d), _lib.ptr(sdf), n,
                                                _lib.ptr(off3), _lib.ptr(dweight), _lib.ptr(dlast), _lib.ptr(grad),
                                                _lib.ptr(dsdf), 0, s), "bwd_rec")
        torch.cuda.synchronize()
        out[cached] = dict(cnt3=cnt3, off3=off3, stats=stats, last=last, hdr=hdr, rec_ray=rec_ray, rec_step=rec_step,
                           rec_w=rec_w, rec_sdf=rec_sdf, grad=grad, dsdf=dsdf)
    a, b = out[False], out[True]
    assert a["hdr"] == b["hdr"]
    for k in ("cnt3", "off3", "stats", "last", "rec_ray", "rec_step", "rec_w", "rec_sdf"):
        assert torch.equal(a[k], b[k]), k                              # bit-identical records
    valid = a["rec_ray"] >= 0
    assert int(valid.sum()) == a["hdr"][0] + a["hdr"][1] > 0
    # backward: the per-record value taps are single stores (exact), the scattered remainder float atomics
    assert torch.equal(a["dsdf"][valid], b["dsdf"][valid])
    scale = float(a["grad"].abs().max().clamp_min(1e-30))
    assert float((a["grad"] - b["grad"]).abs().max()) <= 1e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 31, 1000, 2049, 8192, 25600, 70001])
def test_plan_in_two_launches_equals_the_one_launch_plan(n):
    """esr_fine_plan_totals (the counts the host reads back: many workgroups) + esr_fine_plan_offsets (the scan) against
    esr_fine_plan on random per-ray counts and emissive modes: same offsets, same header (tiles as the host derives them),
    including the registered range flag in bit 1 of the overflow word and a pre-set bit 0."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    L = _lib.lib()
    dev = "cuda:0"
    s = _lib.stream_ptr(dev)
    eng = FineEngine(dev)                     # (registers the device's range flag)
    g = torch.Generator().manual_seed(n)
    cnt3 = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(dev)
    em = torch.randint(0, 3, (n,), generator=g, dtype=torch.int64).to(dev)
    stats = torch.randint(0, 200, (3 * n,), generator=g, dtype=torch.int32).to(dev)
    for flag in (0, 1):
        if eng.range_flag is not None:
            eng.range_flag.fill_(flag)
        off_a, off_b = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        plan_a, plan_b = torch.zeros(8, dtype=torch.int32, device=dev), torch.zeros(8, dtype=torch.int32, device=dev)
        plan_a[7] = plan_b[7] = 1             # (as the march leaves it after an overflow)
        _lib.check(L.esr_fine_plan(_lib.ptr(cnt3), _lib.ptr(em), _lib.ptr(stats), n, _lib.ptr(off_a), _lib.ptr(plan_a), s), "plan")
        _lib.check(L.esr_fine_plan_totals(_lib.ptr(cnt3), _lib.ptr(em), _lib.ptr(stats), n, _lib.ptr(plan_b), s), "totals")
        torch.cuda.synchronize()
        hdr = plan_b.tolist()                 # what the host sees between the two launches
        _lib.check(L.esr_fine_plan_offsets(_lib.ptr(cnt3), _lib.ptr(em), n, _lib.ptr(off_b), _lib.ptr(plan_b), s), "offsets")
        torch.cuda.synchronize()
        ref = plan_a.tolist()
        assert hdr[0:2] == ref[0:2] and hdr[4:8] == ref[4:8], (hdr, ref)
        assert [(hdr[0] + 31) // 32, (hdr[0] + 31) // 32 + (hdr[1] + 31) // 32] == ref[2:4]
        assert plan_b.tolist() == ref
        assert torch.equal(off_a, off_b)
        on = em.cpu() == 1
        assert ref[0] == int(cnt3.cpu()[on].sum()) and ref[1] == int(cnt3.cpu()[~on].sum())
        assert ref[7] == 1 | (2 if (flag and eng.range_flag is not None) else 0)
    if eng.range_flag is not None:
        eng.range_flag.zero_()
