"""LTS-stage kernels (include/esr_hip.h section C) vs the CPU oracle (oracle/lts_path.py), forward
and backward (torch autograd through the oracle functions), through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _scene_and_consts(name="tiny"):
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import analytic_sdf, slab_scene
    from oracle import fine_path as fp
    sc = slab_scene(name)
    cfg = lts_cfg("cpu")
    c = fp.make_consts(cfg.app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
                       sc.mask_density, sc.near, sc.num_voxels)
    g = torch.Generator().manual_seed(1)
    grid = analytic_sdf(c.world_size.tolist(), sc.xyz_min, sc.xyz_max) + 0.02 * torch.randn(
        1, 1, *c.world_size.tolist(), generator=g)
    scene = make_scene(sc.xyz_min.tolist(), sc.xyz_max.tolist(), sc.xyz_min.tolist(), sc.xyz_max.tolist(),
                       c.world_size.tolist(), [32, 32, 32], sc.near, float(c.stepsize * c.voxel_size),
                       float(c.voxel_size), c.act_shift, c.maskcache_thres, c.fastcolor_thres, 20.0,
                       c.grad_feat.tolist())
    return sc, c, grid, scene


@pytest.mark.parametrize("with_noise", [False, True])
def test_expgrad_fwd_bwd(with_noise):
    from esr_nerf_amd import _lib
    from oracle import lts_path as lp
    L = _lib.lib()
    sc, c, grid, scene = _scene_and_consts()
    g = torch.Generator().manual_seed(2)
    n = 5000
    pts = sc.xyz_min + (sc.xyz_max - sc.xyz_min) * torch.rand(n, 3, generator=g)
    pts[:50] = sc.xyz_max                      # exactly on the upper faces
    pts[50:100, 0] = sc.xyz_min[0]
    noise = torch.randn(n, 3, generator=g) if with_noise else None
    eps = 0.003
    grid_r = grid.clone().requires_grad_()
    q = pts + noise * eps if with_noise else pts
    q = torch.maximum(torch.minimum(q, sc.xyz_max), sc.xyz_min)
    if with_noise:                                   # keep the perturbed points inside the box
        noise = (q - pts) / eps
    sdf, gr = lp.sdf_expgrad(c, grid_r, q)
    gout = torch.randn(n, 4, generator=g)
    (sdf * gout[:, 0]).sum().add((gr * gout[:, 1:]).sum()).backward()
    out = torch.empty(n, 4, device="cuda")
    s = _lib.stream_ptr("cuda:0")
    gd = grid[0, 0].contiguous().cuda()
    nz = noise.cuda().contiguous() if with_noise else None
    pd = pts.cuda().contiguous()
    _lib.check(L.esr_expgrad_fwd(C.byref(scene), None, None, None, None, _lib.ptr(pd), _lib.ptr(nz), C.c_float(eps),
                                 _lib.ptr(gd), n, 0, _lib.ptr(out), s), "expgrad_fwd")
    assert rel_err(out[:, 0], sdf.detach()) < 1e-5
    assert rel_err(out[:, 1:], gr.detach()) < 1e-5
    gs = torch.zeros_like(gd)
    gout_d = gout.cuda().contiguous()
    _lib.check(L.esr_expgrad_bwd(C.byref(scene), None, None, None, None, _lib.ptr(pd), _lib.ptr(nz), C.c_float(eps),
                                 _lib.ptr(gout_d), n, 0, _lib.ptr(gs), s), "expgrad_bwd")
    assert rel_err(gs, grid_r.grad[0, 0]) < 1e-5


def test_lts_dirs():
    from esr_nerf_amd import _lib
    from oracle import lts_path as lp
    g = torch.Generator().manual_seed(3)
    P, R1 = 37, 65
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    raw = torch.randn(P, R1, 3, generator=g)
    ref = lp.hemisphere_dirs(nrm, raw)
    out = torch.empty(P, R1, 3, device="cuda")
    raw_d, nrm_d = raw.cuda(), nrm.cuda()            # keep the device copies alive across the async launch
    _lib.check(_lib.lib().esr_lts_dirs(_lib.ptr(raw_d), _lib.ptr(nrm_d), P, R1, _lib.ptr(out),
                                       _lib.stream_ptr("cuda:0")), "dirs")
    assert rel_err(out, ref) < 1e-6
    assert bool(((out.cpu() * nrm[:, None]).sum(-1) >= -1e-7).all())


@pytest.mark.parametrize("pdra,P,R", [(0, 10, 8), (1, 10, 8), (0, 100, 256), (1, 7, 300)])
def test_lts_combine_fwd_bwd(pdra, P, R):
    """env map + Disney reflection + hemisphere means + emo_hat assembly, against the oracle formulas."""
    from esr_nerf_amd import _lib
    from oracle import lts_path as lp
    L = _lib.lib()
    g = torch.Generator().manual_seed(10 + P + R)
    rnd = lambda *s: torch.rand(*s, generator=g)
    base, rough, metal = rnd(P, 3).requires_grad_(), rnd(P, 1).requires_grad_(), rnd(P, 1).requires_grad_()
    if P > 5:
        with torch.no_grad():
            rough[0] = 1e-5       # exercises the r^2 clamp
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    view = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    dirs_all = lp.hemisphere_dirs(nrm, torch.randn(P, R + 1, 3, generator=g))
    off_m = (rnd(P * R, 3) * 2).requires_grad_()
    emo_m = (rnd(P * R, 3) * 2).requires_grad_()
    last2 = rnd(P * R).requires_grad_()
    emission = rnd(P, 3).requires_grad_()
    umask = (torch.arange(P) % 3 == 0)
    J = 48
    Pm = {"envmap.mus": (torch.randn(J, 3, generator=g) * 0.3).requires_grad_(),
          "envmap.lambdas": (10 + 20 * torch.randn(J, 1, generator=g)).requires_grad_(),
          "envmap.lobes": torch.randn(J, 3, generator=g).requires_grad_()}
    # oracle, with the reference's tensor plumbing (esrnerf.py:553-572,653-677)
    rep = lambda t: t.repeat([2] + [1] * (t.dim() - 1))
    ex = lambda t: t.view(P, 1, -1).expand(P, R, t.shape[-1]).flatten(0, 1)
    d2 = dirs_all[:, :-1].flatten(0, 1)
    v_rand = -dirs_all[:, -1]
    Rf = lp.disney_reflection(rep(ex(base)), rep(ex(rough)), rep(ex(metal)), rep(ex(nrm)), rep(d2),
                              torch.cat([-ex(view), -ex(v_rand)], 0))
    env = lp.sg_envmap(Pm, d2) * last2.unsqueeze(-1)
    off_hat = (rep(off_m + env) * Rf).view(-1, R, 3).mean(-2)
    reflect = (rep(emo_m) * Rf).view(-1, R, 3).mean(-2)
    if pdra:
        um = rep(umask)
        emo_hat = torch.where(um[:, None], rep(emission) + reflect.detach(), reflect)
    else:
        emo_hat = rep(emission) + reflect
    g1, g2 = torch.randn(2 * P, 3, generator=g), torch.randn(2 * P, 3, generator=g)
    ((off_hat * g1).sum() + (emo_hat * g2).sum()).backward()

    dev = lambda t: t.detach().cuda().contiguous()
    keep = dict(base=dev(base), rough=dev(rough[:, 0]), metal=dev(metal[:, 0]), normal=dev(nrm), view=dev(view),
                dirs=dev(dirs_all), off_m=dev(off_m), emo_m=dev(emo_m), last2=dev(last2), mus=dev(Pm["envmap.mus"]),
                lambdas=dev(Pm["envmap.lambdas"][:, 0]), lobes=dev(Pm["envmap.lobes"]), emission=dev(emission),
                umask=umask.to(torch.uint8).cuda())
    a = _lib.EsrLtsArgs()
    a.n_pts, a.n_rays, a.n_sg, a.pdra_mode = P, R, J, pdra
    for k, v in keep.items():
        setattr(a, k, v.data_ptr())
    oh = torch.empty(2 * P, 3, device="cuda")
    eh = torch.empty(2 * P, 3, device="cuda")
    s = _lib.stream_ptr("cuda:0")
    _lib.check(L.esr_lts_combine_fwd(C.byref(a), _lib.ptr(oh), _lib.ptr(eh), s), "combine_fwd")
    assert rel_err(oh, off_hat.detach()) < 2e-5
    assert rel_err(eh, emo_hat.detach()) < 2e-5
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    gr = dict(d_off_m=z(P * R, 3), d_emo_m=z(P * R, 3), d_last2=z(P * R), d_base=z(P, 3), d_rough=z(P), d_metal=z(P),
              d_emission=z(P, 3), d_mus=z(J, 3), d_lambdas=z(J), d_lobes=z(J, 3))
    gs = _lib.EsrLtsGrads()
    for k, v in gr.items():
        setattr(gs, k, v.data_ptr())
    g1d, g2d = g1.cuda(), g2.cuda()
    _lib.check(L.esr_lts_combine_bwd(C.byref(a), _lib.ptr(g1d), _lib.ptr(g2d), C.byref(gs), s), "combine_bwd")
    zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    exp = dict(d_off_m=off_m.grad, d_emo_m=zero(emo_m), d_last2=last2.grad, d_base=base.grad, d_rough=rough.grad[:, 0],
               d_metal=metal.grad[:, 0], d_emission=zero(emission), d_mus=Pm["envmap.mus"].grad,
               d_lambdas=Pm["envmap.lambdas"].grad[:, 0], d_lobes=Pm["envmap.lobes"].grad)
    bad = {k: rel_err(gr[k], v) for k, v in exp.items() if not rel_err(gr[k], v) < 1e-4}
    assert not bad, bad


def test_feat_kernels_explicit_points_straddling_the_box():
    """esr_fine_feat_fwd / _bwd in explicit-point mode with points up to ~1 voxel OUTSIDE the box (the
    perturbed emit/brdf re-evaluation of esrnerf.py:807-830 produces them): every tap coordinate is clamped to
    the grid (voxurff.py:697-699), colour taps are zero-padded.  Forward rows and the scatter into the SDF /
    colour gradients against the oracle's stencil."""
    import ctypes as C
    from esr_nerf_amd import _lib
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    L = _lib.lib()
    sc = slab_scene("tiny", s_val=40.0)
    cfg = lts_cfg("cpu")
    c = fp.make_consts(cfg.app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
                       sc.mask_density, sc.near, sc.num_voxels)
    ws = [int(v) for v in c.world_size]
    g = torch.Generator().manual_seed(5)
    sdf = torch.randn(1, 1, *ws, generator=g).requires_grad_(True)
    col = (torch.randn(1, 6, *ws, generator=g) * 0.3).requires_grad_(True)
    n = 200
    lo, hi = sc.xyz_min, sc.xyz_max
    vox = float(c.voxel_size)
    pts = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    face = torch.randint(0, 3, (n,), generator=g)
    side = torch.randint(0, 2, (n,), generator=g).bool()
    off = (torch.rand(n, generator=g) * 2 - 1) * 0.45 * vox             # +-0.45 voxel around a face
    for i in range(n // 2):                                             # half of the points hug a face
        a = int(face[i])
        pts[i, a] = (hi[a] if side[i] else lo[a]) + off[i]
    pts[0] = lo - 0.3 * vox                                             # outside on all three axes (corner)
    pts[1] = hi + 0.3 * vox
    pts[2] = lo - 1.2 * vox      # beyond the smallest tap radius: +- taps clamp onto each other (diff = 0 -> the
    pts[3, 0] = hi[0] + 0.8 * vox  # 1e-12 guard); forward only, the reference's own gradient is 0/1e-12 noise there
    sdfv = torch.randn(n, generator=g)
    feat, _, nrm = fp.sdf_stencil(c, sdf, pts, c.grad_feat, diff_eps=1e-12)
    colv = fp.sample_grid(col, fp.to_norm(pts, lo, hi))
    scene = make_scene(lo.tolist(), hi.tolist(), lo.tolist(), hi.tolist(), ws, [32, 32, 32], sc.near,
                       float(c.stepsize * c.voxel_size), vox, 0.0, 1e-3, 1e-4, 40.0, [float(v) for v in c.grad_feat])
    tiles = (n + 31) // 32
    dev = "cuda"
    pd, vd, sv = pts.cuda().contiguous(), torch.zeros(n, 3, device=dev), sdfv.cuda().contiguous()
    sdf_d = sdf.detach()[0, 0].contiguous().cuda()
    col_d = col.detach()[0].permute(1, 2, 3, 0).contiguous().cuda()
    fa = _lib.EsrFeatArgs()
    fa.pts, fa.pt_viewdirs, fa.pt_sdf, fa.n_pts = pd.data_ptr(), vd.data_ptr(), sv.data_ptr(), n
    fa.sdf = sdf_d.data_ptr()
    fa.color_off[0] = col_d.data_ptr()
    fa.tiles_on, fa.tiles_all = 0, tiles
    X = torch.empty(tiles * 104 * 32, device=dev)
    gn = torch.empty(tiles * 4 * 32, device=dev)
    s = _lib.stream_ptr("cuda:0")
    _lib.check(L.esr_fine_feat_fwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), s), "feat_fwd")
    Xr = X.view(tiles, 104, 32).permute(0, 2, 1).reshape(tiles * 32, 104)[:n].cpu()
    assert rel_err(Xr[:, 0:6], colv) < 1e-5
    assert torch.equal(Xr[:, 6], sdfv)
    assert rel_err(Xr[:, 7:31], feat) < 1e-5
    assert rel_err(Xr[:, 31:43], nrm) < 1e-4, rel_err(Xr[:, 31:43], nrm)
    # backward: random dX on the 43 grid-fed rows
    dXr = torch.randn(n, 43, generator=g)
    dXr[2:4] = 0.0
    (colv * dXr[:, 0:6]).sum().add((feat * dXr[:, 7:31]).sum()).add((nrm * dXr[:, 31:43]).sum()).backward()
    dXt = torch.zeros(tiles * 32, 64)
    dXt[:n, :43] = dXr
    dX = dXt.view(tiles, 32, 64).permute(0, 2, 1).contiguous().cuda()
    g_sdf, g_col = torch.zeros_like(sdf_d), torch.zeros_like(col_d)
    dsdf_out = torch.zeros(tiles * 32, device=dev)
    src = (_lib.EsrFeatBwdSrc * 1)()
    src[0].dX, src[0].grad_color_on, src[0].grad_color_off = dX.data_ptr(), None, g_col.data_ptr()
    src[0].t0, src[0].t1 = 0, tiles
    _lib.check(L.esr_fine_feat_bwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), src, 1, None,
                                   _lib.ptr(g_sdf), _lib.ptr(dsdf_out), None, 0, s), "feat_bwd")
    assert rel_err(g_sdf, sdf.grad[0, 0]) < 1e-4, rel_err(g_sdf, sdf.grad[0, 0])
    assert rel_err(g_col.permute(3, 0, 1, 2), col.grad[0]) < 2e-5
    assert rel_err(dsdf_out[:n], dXr[:, 6]) < 1e-6


@pytest.mark.parametrize("zero_pad", [0, 1])
def test_feat_bwd_grad4_equals_expgrad_bwd(zero_pad):
    """esr_fine_feat_bwd's folded exact-gradient scatter (grad4 / grad4_mode) against esr_expgrad_bwd at the same points,
    incl. points up to ~1 voxel outside the box (border-replicated or dropped corners) and tiles whose windows do not all
    fit (random points: one segment each); bit 1 of grad4_mode: the points' own SDF-value gradient as the value component."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    L = _lib.lib()
    sc = slab_scene("tiny", s_val=40.0)
    c = fp.make_consts(lts_cfg("cpu").app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.near, sc.num_voxels)
    ws = [int(v) for v in c.world_size]
    g = torch.Generator().manual_seed(11 + zero_pad)
    n = 300
    lo, hi = sc.xyz_min, sc.xyz_max
    vox = float(c.voxel_size)
    pts = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    pts[:100] = pts[:1] + torch.cumsum(torch.full((100, 3), 0.4 * vox), 0)           # a diagonal run (windows that fit)
    pts[100:140, 0] = hi[0] + (torch.rand(40, generator=g) * 2 - 1) * 0.9 * vox        # around / beyond a face
    pts[140:150] = lo - 0.7 * vox
    scene = make_scene(lo.tolist(), hi.tolist(), lo.tolist(), hi.tolist(), ws, [32, 32, 32], sc.near,
                       float(c.stepsize * c.voxel_size), vox, 0.0, 1e-3, 1e-4, 40.0, [float(v) for v in c.grad_feat])
    tiles = (n + 31) // 32
    dev = "cuda"
    pd, vd, sv = pts.cuda().contiguous(), torch.zeros(n, 3, device=dev), torch.zeros(n, device=dev)
    sdf_d = torch.randn(*ws, generator=g).cuda()
    fa = _lib.EsrFeatArgs()
    fa.pts, fa.pt_viewdirs, fa.pt_sdf, fa.n_pts = pd.data_ptr(), vd.data_ptr(), sv.data_ptr(), n
    fa.sdf = sdf_d.data_ptr()
    fa.tiles_on, fa.tiles_all = 0, tiles
    X = torch.empty(tiles * 104 * 32, device=dev)
    gn = torch.empty(tiles * 4 * 32, device=dev)
    s = _lib.stream_ptr("cuda:0")
    _lib.check(L.esr_fine_feat_fwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), s), "feat_fwd")
    g4 = torch.zeros(tiles * 32, 4, device=dev)
    g4[:n] = torch.randn(n, 4, generator=g).cuda()
    dX = torch.zeros(tiles * 64 * 32, device=dev)
    src = (_lib.EsrFeatBwdSrc * 1)()
    src[0].dX, src[0].t0, src[0].t1 = dX.data_ptr(), 0, tiles
    # (a) grad4 alone (all dX rows zero: the stencil contributes nothing)
    ours, ref = torch.zeros_like(sdf_d), torch.zeros_like(sdf_d)
    dsdf_out = torch.zeros(tiles * 32, device=dev)
    _lib.check(L.esr_fine_feat_bwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), src, 1, None, _lib.ptr(ours),
                                   _lib.ptr(dsdf_out), _lib.ptr(g4), zero_pad, s), "feat_bwd")
    _lib.check(L.esr_expgrad_bwd(C.byref(scene), None, None, None, None, _lib.ptr(pd), None, C.c_float(0.0), _lib.ptr(g4), n,
                                 zero_pad, _lib.ptr(ref), s), "expgrad_bwd")
    assert float(ref.abs().max()) > 0.1
    assert rel_err(ours, ref) < 1e-5, rel_err(ours, ref)
    # (b) the points' own SDF-value gradient (dX row 6) scattered as the value component
    dXt = torch.zeros(tiles * 32, 64)
    dXt[:n, 6] = torch.randn(n, generator=g)
    dX2 = dXt.view(tiles, 32, 64).permute(0, 2, 1).contiguous().cuda()
    src[0].dX = dX2.data_ptr()
    ours2, ref2 = torch.zeros_like(sdf_d), torch.zeros_like(sdf_d)
    _lib.check(L.esr_fine_feat_bwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), src, 1, None, _lib.ptr(ours2),
                                   None, None, 2 | zero_pad, s), "feat_bwd")
    g4v = torch.zeros(tiles * 32, 4, device=dev)
    g4v[:, 0] = dXt[:, 6].cuda()
    _lib.check(L.esr_expgrad_bwd(C.byref(scene), None, None, None, None, _lib.ptr(pd), None, C.c_float(0.0), _lib.ptr(g4v), n,
                                 zero_pad, _lib.ptr(ref2), s), "expgrad_bwd")
    assert rel_err(ours2, ref2) < 1e-5, rel_err(ours2, ref2)


def test_feat_bwd_many_short_ray_pieces_per_tile():
    """The feature backward's per-segment LDS windows on tiles that hold pieces of up to ~20 rays, with padding lanes in
    the middle of tiles (the LTS stages' secondary pass looks like this; round 2's form handled four rays per tile in
    windows and sent the rest to global atomics): march-record mode, positions from esr_sample_points, gradients against
    autograd through the oracle's stencil at those positions."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    L = _lib.lib()
    sc = slab_scene("tiny", s_val=40.0)
    c = fp.make_consts(lts_cfg("cpu").app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.near, sc.num_voxels)
    ws = [int(v) for v in c.world_size]
    g = torch.Generator().manual_seed(23)
    lo, hi = sc.xyz_min, sc.xyz_max
    vox = float(c.voxel_size)
    stepdist = float(c.stepsize * c.voxel_size)
    n_rays = 150
    # rays from inside the box in random directions (every step stays a few voxels from the start)
    rays_o = lo + (hi - lo) * (0.2 + 0.6 * torch.rand(n_rays, 3, generator=g))
    rays_d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    rec_ray, rec_step = [], []
    for r in range(n_rays):
        k = int(torch.randint(1, 7, (1,), generator=g))
        first = int(torch.randint(0, 4, (1,), generator=g))
        for q in range(k):
            rec_ray.append(r); rec_step.append(first + q)
        if r % 11 == 0:                                   # a padding lane in the middle of the records
            rec_ray.append(-1); rec_step.append(0)
    tiles = (len(rec_ray) + 31) // 32
    rec_ray += [-1] * (tiles * 32 - len(rec_ray)); rec_step += [0] * (tiles * 32 - len(rec_step))
    rr = torch.tensor(rec_ray, dtype=torch.int32); rs = torch.tensor(rec_step, dtype=torch.int32)
    scene = make_scene(lo.tolist(), hi.tolist(), lo.tolist(), hi.tolist(), ws, [32, 32, 32], 0.0, stepdist, vox, 0.0, 1e-3,
                       1e-4, 40.0, [float(v) for v in c.grad_feat])
    dev = "cuda"
    ro, rd = rays_o.cuda().contiguous(), rays_d.cuda().contiguous()
    rrd, rsd = rr.cuda(), rs.cuda()
    s = _lib.stream_ptr("cuda:0")
    pts = torch.zeros(tiles * 32, 3, device=dev)
    _lib.check(L.esr_sample_points(C.byref(scene), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(rrd), _lib.ptr(rsd), tiles * 32,
                                   _lib.ptr(pts), s), "sample_points")
    live = rr >= 0
    p_live = pts.cpu()[live]
    assert bool(((p_live >= lo) & (p_live <= hi)).all())
    assert max(int((rr[t * 32:(t + 1) * 32][live[t * 32:(t + 1) * 32]]).unique().numel()) for t in range(tiles)) >= 8
    sdf = torch.randn(1, 1, *ws, generator=g).requires_grad_(True)
    col = (torch.randn(1, 6, *ws, generator=g) * 0.3).requires_grad_(True)
    feat, _, nrm = fp.sdf_stencil(c, sdf, p_live, c.grad_feat, diff_eps=1e-12)
    colv = fp.sample_grid(col, fp.to_norm(p_live, lo, hi))
    sdf_d = sdf.detach()[0, 0].contiguous().cuda()
    col_d = col.detach()[0].permute(1, 2, 3, 0).contiguous().cuda()
    rsdf = torch.zeros(tiles * 32, device=dev)
    fa = _lib.EsrFeatArgs()
    fa.rays_o, fa.rays_d, fa.viewdirs = ro.data_ptr(), rd.data_ptr(), rd.data_ptr()
    fa.rec_ray, fa.rec_step, fa.rec_sdf = rrd.data_ptr(), rsd.data_ptr(), rsdf.data_ptr()
    fa.sdf = sdf_d.data_ptr()
    fa.color_off[0] = col_d.data_ptr()
    fa.tiles_on, fa.tiles_all = 0, tiles
    X = torch.empty(tiles * 104 * 32, device=dev)
    gn = torch.empty(tiles * 4 * 32, device=dev)
    _lib.check(L.esr_fine_feat_fwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), s), "feat_fwd")
    Xr = X.view(tiles, 104, 32).permute(0, 2, 1).reshape(tiles * 32, 104).cpu()[live]
    assert rel_err(Xr[:, 7:31], feat) < 1e-5 and rel_err(Xr[:, 31:43], nrm) < 1e-4
    dXr = torch.randn(int(live.sum()), 43, generator=g)
    dXr[:, 6] = 0.0                                       # (record mode: the SDF value row is a grid tap of the march)
    (colv * dXr[:, 0:6]).sum().add((feat * dXr[:, 7:31]).sum()).add((nrm * dXr[:, 31:43]).sum()).backward()
    dXt = torch.zeros(tiles * 32, 64)
    dXt[live, :43] = dXr
    dX = dXt.view(tiles, 32, 64).permute(0, 2, 1).contiguous().cuda()
    g_sdf, g_col = torch.zeros_like(sdf_d), torch.zeros_like(col_d)
    src = (_lib.EsrFeatBwdSrc * 1)()
    src[0].dX, src[0].grad_color_on, src[0].grad_color_off = dX.data_ptr(), None, g_col.data_ptr()
    src[0].t0, src[0].t1 = 0, tiles
    _lib.check(L.esr_fine_feat_bwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), src, 1, None,
                                   _lib.ptr(g_sdf), None, None, 0, s), "feat_bwd")
    assert rel_err(g_sdf, sdf.grad[0, 0]) < 1e-4, rel_err(g_sdf, sdf.grad[0, 0])
    assert rel_err(g_col.permute(3, 0, 1, 2), col.grad[0]) < 2e-5


def test_feat_fwd_x16_tile_is_the_fp32_tile_rounded_to_bf16():
    """esr_fine_feat_fwd_x16: every row of the bf16 row-quad tile equals the fp32 tile's row rounded to bf16 (what the bf16
    kernels made of it on load), the two alternate quads carry the colour group of rows 88..93 in front of rows 6, 7, padding
    lanes are zero, and the fp32 tile still receives the normal rows."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    L = _lib.lib()
    sc = slab_scene("tiny", s_val=40.0)
    c = fp.make_consts(lts_cfg("cpu").app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.near, sc.num_voxels)
    ws = [int(v) for v in c.world_size]
    g = torch.Generator().manual_seed(31)
    n = 150                                                     # 5 tiles, the last one with 22 padding lanes
    lo, hi = sc.xyz_min, sc.xyz_max
    pts = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    vox = float(c.voxel_size)
    scene = make_scene(lo.tolist(), hi.tolist(), lo.tolist(), hi.tolist(), ws, [32, 32, 32], sc.near,
                       float(c.stepsize * c.voxel_size), vox, 0.0, 1e-3, 1e-4, 40.0, [float(v) for v in c.grad_feat])
    tiles = (n + 31) // 32
    dev = "cuda"
    pd = pts.cuda().contiguous()
    vd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda().contiguous()
    sv = torch.randn(n, generator=g).cuda()
    sdf_d = torch.randn(*ws, generator=g).cuda()
    col_a = (torch.randn(*ws, 6, generator=g) * 0.3).cuda().contiguous()
    col_b = (torch.randn(*ws, 6, generator=g) * 0.3).cuda().contiguous()
    fa = _lib.EsrFeatArgs()
    fa.pts, fa.pt_viewdirs, fa.pt_sdf, fa.n_pts = pd.data_ptr(), vd.data_ptr(), sv.data_ptr(), n
    fa.sdf = sdf_d.data_ptr()
    fa.color_on[0], fa.color_on[1] = col_a.data_ptr(), col_b.data_ptr()
    fa.color_off[0] = col_b.data_ptr()
    fa.tiles_on, fa.tiles_all = 2, tiles                         # two emissive-on tiles (both colour groups), three others
    s = _lib.stream_ptr("cuda:0")
    X = torch.empty(tiles * 104 * 32, device=dev)
    gn = torch.empty(tiles * 4 * 32, device=dev)
    _lib.check(L.esr_fine_feat_fwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), s), "feat_fwd")
    assert int(L.esr_fine_feat_x16_bytes(tiles)) == tiles * 26 * 256
    X16 = torch.full((tiles * 26 * 256,), 0x7f, dtype=torch.uint8, device=dev)
    Xn = torch.full_like(X, float("nan"))
    gn2 = torch.empty_like(gn)
    _lib.check(L.esr_fine_feat_fwd_x16(C.byref(scene), C.byref(fa), _lib.ptr(Xn), _lib.ptr(gn2), _lib.ptr(X16), s), "feat_fwd_x16")
    Xr = X.view(tiles, 104, 32).cpu()
    want = Xr.to(torch.bfloat16)                                 # [tile][row][sample]
    q = X16.cpu().view(torch.bfloat16).view(tiles, 26, 32, 4)    # [tile][quad][slot][row in quad]
    sidx = torch.arange(32)
    slot = 8 * ((sidx >> 1) & 3) + 2 * (sidx >> 3) + (sidx & 1)
    got = q[:, :24][:, :, slot, :].permute(0, 1, 3, 2).reshape(tiles, 96, 32)     # -> [tile][row][sample]
    assert torch.equal(got.view(torch.int16), want[:, :96].contiguous().view(torch.int16))
    alt = q[:, 24:26][:, :, slot, :].permute(0, 1, 3, 2).reshape(tiles, 8, 32)
    want_alt = torch.cat([want[:, 88:94], want[:, 6:8]], 1)
    assert torch.equal(alt.view(torch.int16), want_alt.contiguous().view(torch.int16))
    assert bool((want[:2, 88:94].float().abs() > 0).any())       # (the second colour group is fed on the emissive-on tiles)
    assert torch.equal(gn2, gn)
    Xn = Xn.view(tiles, 104, 32).cpu()
    assert torch.equal(Xn[:, 31:43], Xr[:, 31:43])               # the normal rows, fp32 (the backward reads them)
    assert bool(torch.isnan(Xn[:, :31]).all()) and bool(torch.isnan(Xn[:, 43:]).all())


def test_feat_fwd_direct_form_for_wide_stencils():
    """cfg grad_feat radii beyond 2 voxels (voxurff.py:164-167 takes any list): the forward's stencil bars do not reach,
    the direct form runs (feat.hip: feat_fwd_kernel<false>) -- rows against the oracle's stencil; the scatter, whose bars
    and LDS windows are sized for radii <= 2, refuses (ESR_ECAP) instead of dropping the outer taps."""
    import ctypes as C
    from esr_nerf_amd import _lib
    from esr_nerf_amd.config import lts_cfg
    from esr_nerf_amd.fine_engine import make_scene
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    L = _lib.lib()
    sc = slab_scene("tiny", s_val=40.0)
    c = fp.make_consts(lts_cfg("cpu").app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                       sc.mask_alpha_init, sc.mask_density, sc.near, sc.num_voxels)
    ws = [int(v) for v in c.world_size]
    g = torch.Generator().manual_seed(9)
    sdf = torch.randn(1, 1, *ws, generator=g)
    n = 160
    lo, hi = sc.xyz_min, sc.xyz_max
    pts = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    pts[:40, 2] = lo[2] + (hi[2] - lo[2]) * 0.02 * torch.rand(40, generator=g)       # near a face: the wide taps clamp
    radii = torch.tensor([0.5, 1.0, 2.5, 3.0])
    feat, _, nrm = fp.sdf_stencil(c, sdf, pts, radii, diff_eps=1e-12)
    vox = float(c.voxel_size)
    tiles = (n + 31) // 32
    pd, vd, sv = pts.cuda().contiguous(), torch.zeros(n, 3, device="cuda"), torch.zeros(n, device="cuda")
    sdf_d = sdf[0, 0].contiguous().cuda()
    fa = _lib.EsrFeatArgs()
    fa.pts, fa.pt_viewdirs, fa.pt_sdf, fa.n_pts = pd.data_ptr(), vd.data_ptr(), sv.data_ptr(), n
    fa.sdf = sdf_d.data_ptr()
    fa.tiles_on, fa.tiles_all = 0, tiles
    X = torch.empty(tiles * 104 * 32, device="cuda")
    gn = torch.empty(tiles * 4 * 32, device="cuda")
    s = _lib.stream_ptr("cuda:0")
    out = {}
    for name, rr in (("wide", radii), ("default", torch.tensor([0.5, 1.0, 1.5, 2.0]))):
        scene = make_scene(lo.tolist(), hi.tolist(), lo.tolist(), hi.tolist(), ws, [32, 32, 32], sc.near,
                           float(c.stepsize * c.voxel_size), vox, 0.0, 1e-3, 1e-4, 40.0, [float(v) for v in rr])
        _lib.check(L.esr_fine_feat_fwd(C.byref(scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), s), "feat_fwd")
        out[name] = X.view(tiles, 104, 32).permute(0, 2, 1).reshape(tiles * 32, 104)[:n].cpu().clone()
        if name == "wide":
            wide_scene = scene
    assert rel_err(out["wide"][:, 7:31], feat) < 1e-5
    assert rel_err(out["wide"][:, 31:43], nrm) < 1e-4
    # radii 0.5 and 1.0 are in both lists: the bar form (default radii) and the direct form agree bit for bit on them
    for ar in range(6):
        for k in (0, 1):
            assert torch.equal(out["wide"][:, 7 + ar * 4 + k], out["default"][:, 7 + ar * 4 + k])
    dX = torch.zeros(tiles * 64 * 32, device="cuda")
    gs = torch.zeros_like(sdf_d)
    src = (_lib.EsrFeatBwdSrc * 1)()
    src[0].dX, src[0].t0, src[0].t1 = dX.data_ptr(), 0, tiles
    rc = L.esr_fine_feat_bwd(C.byref(wide_scene), C.byref(fa), _lib.ptr(X), _lib.ptr(gn), src, 1, None, _lib.ptr(gs),
                             None, None, 0, s)
    assert rc == -2
    with pytest.raises(RuntimeError, match="capacity"):
        _lib.check(rc, "feat_bwd")


def test_emit_edit_kernel_all_modes():
    """esr_emit_edit against the oracle's restatement of esrnerf.py:427-441 + the reference's hsv pair."""
    from esr_nerf_amd import _lib
    from oracle import lts_path as lp
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    n = 5000
    emit = torch.rand(n, 3, generator=g) * 4.0 + 1e-3
    emit[:50] = emit[:50, :1]                                  # grey: delta == 0 branch of rgb_to_hsv
    modes = torch.randint(0, 5, (n,), generator=g)
    inten = torch.rand(n, generator=g) * 3.0
    cols = torch.rand(n, 2, generator=g)
    cols[100:120, 0] = torch.tensor([0.0, 1.0 / 6, 2.0 / 6, 0.5, 4.0 / 6, 5.0 / 6, 0.999999, 1.0, 0.25, 0.75] * 2)
    ref = lp.edit_emission(emit, modes, inten, cols)
    e, m, i, c = emit.cuda().contiguous(), modes.cuda(), inten.cuda(), cols.cuda().contiguous()
    _lib.check(L.esr_emit_edit(_lib.ptr(e), _lib.ptr(m), _lib.ptr(i), _lib.ptr(c), n, _lib.stream_ptr("cuda:0")), "emit_edit")
    assert rel_err(e, ref) < 1e-6
