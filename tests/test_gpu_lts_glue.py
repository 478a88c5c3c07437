"""The fused glue kernels of the light-transport step (csrc/lts.hip: esr_lts_ref_order / perturb / gather_rows /
gather_points) against the torch lines they replace (the reference's esrnerf.py:781-830 spelled with torch ops)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layout(n_rays=37, seed=0):
    """A synthetic record layout as the march leaves it: per-ray counts, on rays first (from slot 0), off rays from the
    next multiple of 32; padding slots -1."""
    g = torch.Generator().manual_seed(seed)
    cnt = torch.randint(0, 9, (n_rays,), generator=g)
    on = torch.rand(n_rays, generator=g) < 0.4
    off = torch.zeros(n_rays, dtype=torch.long)
    pos = 0
    for r in range(n_rays):
        if on[r]:
            off[r] = pos
            pos += int(cnt[r])
    n_on = pos
    t_on = (n_on + 31) // 32
    pos = t_on * 32
    for r in range(n_rays):
        if not on[r]:
            off[r] = pos
            pos += int(cnt[r])
    n_off = pos - t_on * 32
    tiles = t_on + (n_off + 31) // 32
    rec_ray = torch.full((tiles * 32,), -1, dtype=torch.int32)
    for r in range(n_rays):
        rec_ray[off[r]: off[r] + cnt[r]] = r
    return cnt.int(), off.int(), rec_ray, tiles, n_on, n_off, t_on


def test_ref_order_and_perturb_vs_torch():
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr(DEV)
    cnt3, off3, rec_ray, T, n_on, n_off, t_on = _layout()
    m3 = n_on + n_off
    # the torch lines (lts_engine._ref_order before the fusion)
    jidx = torch.cat([torch.arange(n_on), t_on * 32 + torch.arange(n_off)])
    ray_j = rec_ray.long()[jidx]
    ref_off = torch.cumsum(cnt3.long(), 0) - cnt3.long()
    ref_pos = ref_off[ray_j] + (jidx - off3.long()[ray_j])
    perm_ref = torch.empty(m3, dtype=torch.long)
    perm_ref[ref_pos] = jidx
    d = lambda t: t.to(DEV).contiguous()
    cnt_d, off_d, rr_d = d(cnt3), d(off3), d(rec_ray)
    csum = torch.cumsum(cnt_d, 0, dtype=torch.int64)
    perm = torch.full((m3,), -7, dtype=torch.long, device=DEV)
    ray64 = torch.empty(T * 32, dtype=torch.long, device=DEV)
    _lib.check(L.esr_lts_ref_order(_lib.ptr(rr_d), _lib.ptr(cnt_d), _lib.ptr(off_d), _lib.ptr(csum), T * 32, _lib.ptr(perm),
                                   _lib.ptr(ray64), s), "ref_order")
    assert torch.equal(perm.cpu(), perm_ref) and torch.equal(ray64.cpu(), rec_ray.long())
    # perturbed positions / scattered noise
    g = torch.Generator().manual_seed(1)
    pts_all, nn, ne = torch.randn(T * 32, 3, generator=g), torch.randn(m3, 3, generator=g), torch.randn(m3, 3, generator=g)
    eps = 0.0123
    pts_e_ref = pts_all[perm_ref] + ne * eps
    noise_ref = torch.zeros(T * 32, 3)
    noise_ref[perm_ref] = nn
    pa, nnd, ned = d(pts_all), d(nn), d(ne)
    noise = torch.zeros(T * 32, 3, device=DEV)
    pts_e = torch.empty(m3, 3, device=DEV)
    _lib.check(L.esr_lts_perturb(_lib.ptr(pa), _lib.ptr(perm), _lib.ptr(nnd), _lib.ptr(ned), C.c_float(eps), m3,
                                 _lib.ptr(noise), _lib.ptr(pts_e), s), "perturb")
    assert torch.equal(pts_e.cpu(), pts_e_ref) and torch.equal(noise.cpu(), noise_ref)          # same two roundings: bit-exact


@pytest.mark.parametrize("with_perm", [True, False])
def test_gather_rows_vs_torch(with_perm):
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(2)
    T, rows = 9, 8
    tm = torch.randn(T, rows, 32, generator=g)                          # tile-major
    rm = tm.permute(0, 2, 1).reshape(T * 32, rows)                      # its row-major view
    n = 200
    perm = torch.randperm(T * 32, generator=g)[:n] if with_perm else None
    sel = perm if with_perm else torch.arange(n)
    tm_d, rm_d = tm.to(DEV).contiguous(), rm.to(DEV).contiguous()
    perm_d = perm.to(DEV) if with_perm else None
    for src, tile_rows, stride, c0, nch in ((tm_d, rows, 0, 0, 5), (tm_d, rows, 0, 2, 3), (rm_d, 0, rows, 1, 3)):
        out = torch.empty(n, nch, device=DEV)
        _lib.check(L.esr_lts_gather_rows(_lib.ptr(src), tile_rows, stride, c0, nch, _lib.ptr(perm_d), n, _lib.ptr(out), s),
                   "gather_rows")
        assert torch.equal(out.cpu(), rm[sel][:, c0:c0 + nch])
    assert L.esr_lts_gather_rows(_lib.ptr(tm_d), rows, 0, 6, 3, None, n, _lib.ptr(out), s) != 0     # columns past the tile


def test_gather_points_vs_torch():
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(3)
    T, N, P = 7, 40, 33
    ray64 = torch.randint(0, N, (T * 32,), generator=g)
    jp = torch.randperm(T * 32, generator=g)[:P]
    pts_all, eg = torch.randn(T * 32, 3, generator=g), torch.randn(T * 32, 4, generator=g)
    eg[jp[0], 1:] = 0.0                                                  # a zero gradient: F.normalize's eps branch
    rec_sdf, viewdirs = torch.randn(T * 32, generator=g), torch.randn(N, 3, generator=g)
    brdf_a, emit_a = torch.rand(T, 8, 32, generator=g), torch.rand(T, 4, 32, generator=g)
    um = torch.rand(N, generator=g) < 0.5
    d = lambda t: t.to(DEV).contiguous()
    keep = dict(jp=d(jp), ray64=d(ray64), pts_all=d(pts_all), eg=d(eg), rec_sdf=d(rec_sdf), viewdirs=d(viewdirs),
                brdf_a=d(brdf_a), emit_a=d(emit_a), umask_rays=d(um))
    out = dict(pts2=torch.empty(2 * P, 3, device=DEV), vd2=torch.full((2 * P, 3), 9.0, device=DEV),
               sdf2=torch.empty(2 * P, device=DEV), normal=torch.empty(P, 3, device=DEV), base=torch.empty(P, 3, device=DEV),
               rough=torch.empty(P, device=DEV), metal=torch.empty(P, device=DEV), emis=torch.empty(P, 3, device=DEV),
               umask=torch.empty(P, dtype=torch.uint8, device=DEV))
    a = _lib.EsrLtsGather()
    a.n_pts = P
    for k, v in {**keep, **out}.items():
        setattr(a, k, v.data_ptr())
    _lib.check(L.esr_lts_gather_points(C.byref(a), s), "gather_points")
    torch.cuda.synchronize()
    brdf_rm, emit_rm = brdf_a.permute(0, 2, 1).reshape(T * 32, 8), emit_a.permute(0, 2, 1).reshape(T * 32, 4)
    ray_p = ray64[jp]
    assert torch.equal(out["pts2"].cpu(), torch.cat([pts_all[jp], pts_all[jp]]))
    assert torch.equal(out["vd2"][:P].cpu(), viewdirs[ray_p]) and float((out["vd2"][P:] - 9.0).abs().max()) == 0.0
    assert torch.equal(out["sdf2"].cpu(), torch.cat([rec_sdf[jp], rec_sdf[jp]]))
    assert torch.allclose(out["normal"].cpu(), torch.nn.functional.normalize(eg[jp, 1:4], dim=-1), rtol=0, atol=2e-7)
    assert torch.equal(out["base"].cpu(), brdf_rm[jp, 0:3]) and torch.equal(out["rough"].cpu(), brdf_rm[jp, 3])
    assert torch.equal(out["metal"].cpu(), brdf_rm[jp, 4]) and torch.equal(out["emis"].cpu(), emit_rm[jp, 0:3])
    assert torch.equal(out["umask"].cpu(), um[ray_p].to(torch.uint8))
