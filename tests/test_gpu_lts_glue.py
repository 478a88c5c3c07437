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


# ---- round 4: the batched forms (one launch for several jobs) and the gathers folded into the activation's backward -------
def _tm(t, rows):
    """[tiles*32, c] row-major -> tile-major [tiles, rows, 32] (rows >= c, rest zero)."""
    tiles = t.shape[0] // 32
    out = torch.zeros(tiles, rows, 32)
    out[:, : t.shape[1], :] = t.view(tiles, 32, t.shape[1]).permute(0, 2, 1)
    return out


def test_ref_order_inv_is_the_inverse_permutation():
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    cnt3, off3, rec_ray, T, n_on, n_off, t_on = _layout(n_rays=53, seed=4)
    m3 = n_on + n_off
    d = lambda t: t.to(DEV).contiguous()
    cnt_d, off_d, rr_d = d(cnt3), d(off3), d(rec_ray)
    csum = torch.cumsum(cnt_d, 0, dtype=torch.int64)
    perm, perm2 = torch.empty(m3, dtype=torch.long, device=DEV), torch.empty(m3, dtype=torch.long, device=DEV)
    ray64, ray64b = torch.empty(T * 32, dtype=torch.long, device=DEV), torch.empty(T * 32, dtype=torch.long, device=DEV)
    inv = torch.full((T * 32,), 77, dtype=torch.int32, device=DEV)
    _lib.check(L.esr_lts_ref_order(_lib.ptr(rr_d), _lib.ptr(cnt_d), _lib.ptr(off_d), _lib.ptr(csum), T * 32, _lib.ptr(perm),
                                   _lib.ptr(ray64), s), "ref_order")
    _lib.check(L.esr_lts_ref_order_inv(_lib.ptr(rr_d), _lib.ptr(cnt_d), _lib.ptr(off_d), _lib.ptr(csum), T * 32,
                                       _lib.ptr(perm2), _lib.ptr(ray64b), _lib.ptr(inv), s), "ref_order_inv")
    assert torch.equal(perm, perm2) and torch.equal(ray64, ray64b)
    inv_c, perm_c = inv.cpu().long(), perm.cpu()
    assert torch.equal(inv_c[perm_c], torch.arange(m3))                  # inv o perm = id on the survivors
    assert torch.equal(inv_c < 0, rec_ray < 0)                           # padding slots: -1


def test_dirs_rays_equals_dirs_plus_the_torch_lines():
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(5)
    P, R = 19, 12
    raw, normal, pts = torch.randn(P, R + 1, 3, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g)
    normal = torch.nn.functional.normalize(normal, dim=-1)
    d = lambda t: t.to(DEV).contiguous()
    raw_d, n_d, p_d = d(raw), d(normal), d(pts)
    dirs0 = torch.empty(P, R + 1, 3, device=DEV)
    _lib.check(L.esr_lts_dirs(_lib.ptr(raw_d), _lib.ptr(n_d), P, R + 1, _lib.ptr(dirs0), s), "dirs")
    dirs1, o2, d2, vr = (torch.empty(P, R + 1, 3, device=DEV), torch.empty(P * R, 3, device=DEV), torch.empty(P * R, 3, device=DEV),
                         torch.empty(P, 3, device=DEV))
    _lib.check(L.esr_lts_dirs_rays(_lib.ptr(raw_d), _lib.ptr(n_d), _lib.ptr(p_d), P, R + 1, _lib.ptr(dirs1), _lib.ptr(o2),
                                   _lib.ptr(d2), _lib.ptr(vr), s), "dirs_rays")
    assert torch.equal(dirs0, dirs1)
    assert torch.equal(o2, p_d.repeat_interleave(R, 0)) and torch.equal(d2, dirs0[:, :R].reshape(P * R, 3))
    assert torch.equal(vr, -dirs0[:, R])


def test_gather_rows_batch_equals_single_launches():
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(6)
    T = 11
    tm8, tm4 = torch.randn(T, 8, 32, generator=g).to(DEV), torch.randn(T, 4, 32, generator=g).to(DEV)
    rm4 = torch.randn(T * 32, 4, generator=g).to(DEV)
    perm = torch.randperm(T * 32, generator=g)[:250].to(DEV)
    specs = [(tm8, 8, 0, 0, 5, perm, 250), (tm4, 4, 0, 0, 3, None, 100), (rm4, 0, 4, 1, 3, perm, 250), (tm8, 8, 0, 3, 2, None, T * 32)]
    arr = (_lib.EsrGatherJob * len(specs))()
    outs, refs = [], []
    for jb, (src, tr, st, c0, nch, pm, n) in zip(arr, specs):
        o, r = torch.empty(n, nch, device=DEV), torch.empty(n, nch, device=DEV)
        outs.append(o); refs.append(r)
        jb.src, jb.tile_rows, jb.row_stride, jb.col0, jb.n_ch = src.data_ptr(), tr, st, c0, nch
        jb.perm, jb.n, jb.out = (pm.data_ptr() if pm is not None else None), n, o.data_ptr()
        _lib.check(L.esr_lts_gather_rows(_lib.ptr(src), tr, st, c0, nch, _lib.ptr(pm), n, _lib.ptr(r), s), "gather_rows")
    _lib.check(L.esr_lts_gather_rows_batch(arr, len(specs), s), "gather_rows_batch")
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    assert L.esr_lts_gather_rows_batch(arr, 5, s) != 0                    # more jobs than the batch holds


@pytest.mark.parametrize("act,rows,nch", [(0, 4, 3), (1, 8, 5)])
def test_act_batch_forward_and_backward_with_gathers_vs_torch(act, rows, nch):
    """esr_act_batch against esr_act_fwd / esr_act_bwd fed with the gradient tensor the torch glue used to assemble:
    index_put of the reference-order rows through perm, index_add_ of the per-point extras at jp, + a tile-major part."""
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(7 + act)
    cnt3, off3, rec_ray, T, n_on, n_off, t_on = _layout(n_rays=41, seed=8)
    m3 = n_on + n_off
    live = torch.nonzero(rec_ray >= 0).flatten()
    perm = live[torch.randperm(m3, generator=g)]                          # some bijection survivors <- reference order
    inv = torch.full((T * 32,), -1, dtype=torch.int32)
    inv[perm] = torch.arange(m3, dtype=torch.int32)
    P = 9
    jp = perm[torch.randperm(m3, generator=g)[:P]]
    pt1 = torch.zeros(T * 32, dtype=torch.int32)
    pt1[jp] = torch.arange(1, P + 1, dtype=torch.int32)
    z = torch.randn(T, rows, 32, generator=g) * 3
    g_tile = torch.randn(T, rows, 32, generator=g)
    src = torch.randn(m3, nch, generator=g)
    ex = [(torch.randn(P, 3, generator=g), 0)] + ([(torch.randn(P, generator=g), 3), (torch.randn(P, generator=g), 4)] if nch == 5 else [])
    # ---- what the torch glue built
    G = torch.zeros(T * 32, nch)
    G[perm] = src
    add = torch.zeros(P, nch)
    for t, c0 in ex:
        add[:, c0:c0 + (t.shape[1] if t.dim() > 1 else 1)] = t.view(P, -1)
    G.index_add_(0, jp, add)
    g_full = g_tile.clone()
    g_full[:, :nch, :] += _tm(G, rows)[:, :nch, :]
    d = lambda t: t.to(DEV).contiguous()
    z_d, gfull_d = d(z), d(g_full)
    fwd_ref, bwd_ref = torch.empty_like(z_d), torch.empty_like(z_d)
    _lib.check(L.esr_act_fwd(_lib.ptr(z_d), T, rows, nch, act, _lib.ptr(fwd_ref), s), "act_fwd")
    _lib.check(L.esr_act_bwd(_lib.ptr(z_d), _lib.ptr(gfull_d), T, rows, nch, act, _lib.ptr(bwd_ref), s), "act_bwd")
    # ---- the batch: job 0 forward, job 1 backward with every source
    keep = dict(g_tile=d(g_tile), src=d(src), inv=d(inv), pt1=d(pt1), ex=[d(t) for t, _ in ex])
    fwd, bwd = torch.empty_like(z_d), torch.empty_like(z_d)
    arr = (_lib.EsrActJob * 2)()
    for jb, out in zip(arr, (fwd, bwd)):
        jb.z, jb.out, jb.tiles, jb.rows, jb.n_ch, jb.act = z_d.data_ptr(), out.data_ptr(), T, rows, nch, act
    jb = arr[1]
    jb.bwd, jb.g_tile, jb.src, jb.src_c, jb.n_src = 1, keep["g_tile"].data_ptr(), keep["src"].data_ptr(), nch, m3
    jb.inv, jb.pt1 = keep["inv"].data_ptr(), keep["pt1"].data_ptr()
    for e, ((t, c0), td) in enumerate(zip(ex, keep["ex"])):
        jb.ex[e], jb.ex_c[e], jb.ex_col0[e] = td.data_ptr(), (t.shape[1] if t.dim() > 1 else 1), c0
    _lib.check(L.esr_act_batch(arr, 2, s), "act_batch")
    assert torch.equal(fwd, fwd_ref)
    # same products; the gradient sum is formed in another order (tile + row + extra vs the torch glue's): a few ulps
    assert float((bwd - bwd_ref).abs().max()) <= 4e-6 * float(bwd_ref.abs().max())
    # identity gather (inv NULL): rows k < n_src land on slot k
    n_id = 70
    src_id = torch.randn(n_id, nch, generator=g)
    G2 = torch.zeros(T * 32, nch)
    G2[:n_id] = src_id
    g2_d = d(_tm(G2, rows))
    _lib.check(L.esr_act_bwd(_lib.ptr(z_d), _lib.ptr(g2_d), T, rows, nch, act, _lib.ptr(bwd_ref), s), "act_bwd")
    arr1 = (_lib.EsrActJob * 1)()
    sid = d(src_id)
    jb = arr1[0]
    jb.z, jb.out, jb.tiles, jb.rows, jb.n_ch, jb.act, jb.bwd = z_d.data_ptr(), bwd.data_ptr(), T, rows, nch, act, 1
    jb.src, jb.src_c, jb.n_src = sid.data_ptr(), nch, n_id
    _lib.check(L.esr_act_batch(arr1, 1, s), "act_batch")
    assert torch.equal(bwd, bwd_ref)


def test_pair_loss_batch_equals_single_launches():
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(9)
    n = 300
    a, b = [torch.randn(n, 3, generator=g).to(DEV) for _ in range(3)], [torch.randn(n, 3, generator=g).to(DEV) for _ in range(3)]
    mask = (torch.rand(n, generator=g) < 0.4).to(torch.uint8).to(DEV)
    cnt = (mask == 0).sum(dtype=torch.int32).view(1)
    terms = [(a[0], b[0], 0, 0.7, 0.7, 0.7, None, None), (a[1], b[1], 1, 1.3, 0.4, 0.9, None, None),
             (a[2], None, 0, 0.2, 0.2, 0.0, mask, cnt)]
    loss_ref = torch.zeros(1, device=DEV)
    refs = []
    for x, y, kind, wv, wa, wb, m, c in terms:
        ga = torch.empty_like(x)
        gb = torch.empty_like(x) if y is not None else None
        _lib.check(L.esr_pair_loss_fwd_bwd(_lib.ptr(x), _lib.ptr(y), C.c_int64(n), 3, _lib.ptr(m), 0, _lib.ptr(c), kind,
                                           C.c_float(wv), C.c_float(wa), C.c_float(wb), _lib.ptr(loss_ref), _lib.ptr(ga),
                                           _lib.ptr(gb), s), "pair_loss")
        refs.append((ga, gb))
    loss = torch.zeros(1, device=DEV)
    arr = (_lib.EsrPairJob * len(terms))()
    outs = []
    for jb, (x, y, kind, wv, wa, wb, m, c) in zip(arr, terms):
        ga = torch.empty_like(x)
        gb = torch.empty_like(x) if y is not None else None
        outs.append((ga, gb))
        jb.a, jb.b, jb.rows, jb.cols = x.data_ptr(), (y.data_ptr() if y is not None else None), n, 3
        jb.row_mask, jb.mask_value = (m.data_ptr() if m is not None else None), 0
        jb.count_dev, jb.kind, jb.w_value, jb.w_a, jb.w_b = (c.data_ptr() if c is not None else None), kind, wv, wa, wb
        jb.ga, jb.gb = ga.data_ptr(), (gb.data_ptr() if gb is not None else None)
    _lib.check(L.esr_pair_loss_batch(arr, len(terms), _lib.ptr(loss), s), "pair_loss_batch")
    for (ga, gb), (ra, rb) in zip(outs, refs):
        assert torch.equal(ga, ra) and (gb is None or torch.equal(gb, rb))
    assert abs(float(loss) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref))       # (atomic order of the partial sums)


def test_gather_points_writes_the_slot_to_point_map():
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr(DEV)
    g = torch.Generator().manual_seed(10)
    T, N, P = 5, 20, 17
    d = lambda t: t.to(DEV).contiguous()
    jp = torch.randperm(T * 32, generator=g)[:P]
    keep = dict(jp=d(jp), ray64=d(torch.randint(0, N, (T * 32,), generator=g)), pts_all=d(torch.randn(T * 32, 3, generator=g)),
                eg=d(torch.randn(T * 32, 4, generator=g)), rec_sdf=d(torch.randn(T * 32, generator=g)),
                viewdirs=d(torch.randn(N, 3, generator=g)), brdf_a=d(torch.rand(T, 8, 32, generator=g)),
                emit_a=d(torch.rand(T, 4, 32, generator=g)), umask_rays=d(torch.rand(N, generator=g) < 0.5))
    out = dict(pts2=torch.empty(2 * P, 3, device=DEV), vd2=torch.empty(2 * P, 3, device=DEV), sdf2=torch.empty(2 * P, device=DEV),
               normal=torch.empty(P, 3, device=DEV), base=torch.empty(P, 3, device=DEV), rough=torch.empty(P, device=DEV),
               metal=torch.empty(P, device=DEV), emis=torch.empty(P, 3, device=DEV), umask=torch.empty(P, dtype=torch.uint8, device=DEV),
               pt1=torch.zeros(T * 32, dtype=torch.int32, device=DEV))
    a = _lib.EsrLtsGather()
    a.n_pts = P
    for k, v in {**keep, **out}.items():
        setattr(a, k, v.data_ptr())
    _lib.check(L.esr_lts_gather_points(C.byref(a), s), "gather_points")
    want = torch.zeros(T * 32, dtype=torch.int32)
    want[jp] = torch.arange(1, P + 1, dtype=torch.int32)
    assert torch.equal(out["pt1"].cpu(), want)
