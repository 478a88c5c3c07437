"""The f32 engine's radiance forward on the 16-bit matrix cores from split fp16 planes (csrc/mlp_split.hip) against
(a) a DOUBLE-precision torch chain -- the accuracy claim: the same distance as the fp32 MFMA kernel and as torch's own fp32
chain -- and (b) the fp32 MFMA kernel it replaces: outputs, saved hidden tiles, ReLU masks; then the merged three-pass launch
against its fp32 twin, and the trainer step end to end (both forwards feed the SAME fp32 backward kernels)."""
import ctypes as C

import pytest
import torch

from conftest import rel_err
from test_gpu_fine_path import NET, _in_colmap

pytestmark = pytest.mark.gpu


def _net(g, scale_in=1.0):
    dims = [85, 192, 192, 192, 3]
    Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5) for i in range(4)]
    Bs = [(torch.randn(dims[i + 1], generator=g) * 0.1) for i in range(4)]
    return Ws, Bs


def gain_bound(Ws):
    """csrc/mlp.hip: split_gain_kernel -- the largest running product of the layers' column sums of |W| along the input-gradient
    chain (output layer first, first layer excluded), and 1."""
    cum, worst = 1.0, 1.0
    for W in reversed(Ws[1:]):
        cum *= float(W.double().abs().sum(0).max())
        worst = max(worst, cum)
    return worst


def amax_source(dz_max, Ws):
    """What esr_mlp_dgrad_split leaves in `amax`: max |dz| x max(1, G / 16) -- the weight-gradient kernels' scale source."""
    return dz_max * max(1.0, gain_bound(Ws) / 16.0)


def _pack(L, eng, which, Ws, Bs):
    Wd, Bd = [w.cuda().contiguous() for w in Ws], [b.cuda().contiguous() for b in Bs]
    eng.pack(which, 0, Wd, Bd)
    return Wd, Bd


@pytest.mark.parametrize("tiles,crow,save,xscale", [(1, 0, 1, 1.0), (37, 88, 1, 1.0), (300, 96, 2, 1.0), (64, 0, 0, 1.0),
                                                    (41, 0, 1, 40.0), (41, 0, 1, 1e-3)])
def test_split_forward_vs_double_precision_and_vs_the_f32_kernel(tiles, crow, save, xscale):
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    assert eng.split_fwd
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(tiles * 3 + crow)
    Ws, Bs = _net(g)
    _pack(L, eng, "off", Ws, Bs)
    X = torch.randn(tiles, 104, 32, generator=g) * xscale
    X[:, 7:31] *= 5.0                                                  # the stencil features are the large inputs
    rows = [r for r in range(96) if _in_colmap(0, r) >= 0]
    cols = [_in_colmap(0, r) for r in rows]
    src_rows = [r + crow if r < 6 else r for r in rows]
    x_ref = torch.zeros(tiles * 32, 85, dtype=torch.float64)
    x_ref[:, cols] = X[:, src_rows, :].permute(0, 2, 1).reshape(tiles * 32, len(rows)).double()
    h, hs = x_ref, []
    for i in range(4):
        h = torch.nn.functional.linear(h, Ws[i].double(), Bs[i].double())
        if i < 3:
            h = torch.relu(h)
            hs.append(h)
    tm = lambda t, r: t.reshape(tiles, 32, r).permute(0, 2, 1).contiguous()
    Xd = X.cuda().contiguous()

    def run(split):
        H = [torch.full((tiles, 192, 32), -3.0, device="cuda") for _ in range(3)]
        M = [torch.full((tiles, 3, 64), -3, dtype=torch.int32, device="cuda") for _ in range(3)]
        z = torch.full((tiles, 4, 32), 7.0, device="cuda")
        if split:
            rc = L.esr_mlp_fwd_split(0, _lib.ptr(eng.packed["off"]), _lib.ptr(eng.packed_split["off"]), _lib.ptr(Xd), 0, tiles,
                                     _lib.ptr_array(H), _lib.ptr_array(M), save, crow, _lib.ptr(z), s)
        else:
            rc = L.esr_mlp_fwd(0, _lib.ptr(eng.packed["off"]), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H), _lib.ptr_array(M),
                               save, crow, _lib.ptr(z), s)
        _lib.check(rc, "fwd")
        torch.cuda.synchronize()
        return z, H, M
    zs, Hs, Ms = run(True)
    zf, Hf, Mf = run(False)
    # (a) against double precision: the split kernel is as close as the fp32 MFMA kernel (both ~5e-7 of the largest value)
    e_split, e_f32 = rel_err(zs[:, :3], tm(h, 3)), rel_err(zf[:, :3], tm(h, 3))
    print(f"outputs vs double: split {e_split:.2e}, f32 MFMA {e_f32:.2e}")
    assert e_split < 3e-6 and e_split < 4 * e_f32 + 5e-7
    assert float(zs[:, 3].abs().max()) == 0.0
    if save == 1:
        for l in range(3):
            es, ef = rel_err(Hs[l], tm(hs[l], 192)), rel_err(Hf[l], tm(hs[l], 192))
            assert es < 3e-6 and es < 4 * ef + 5e-7, (l, es, ef)
    else:
        assert all(float((Hs[l] + 3.0).abs().max()) == 0.0 for l in range(3))        # nothing written
    # (b) masks: the f32 kernel's format; the two kernels may differ only on units within rounding of zero
    if save:
        for l in range(3):
            diff = (Ms[l] ^ Mf[l])
            nd = int(sum(bin(int(v) & 0xffffffff).count("1") for v in diff.flatten().tolist())) if diff.any() else 0
            assert nd <= max(2, tiles * 32 * 192 // 20000), (l, nd)
    else:
        assert all(int((Ms[l] + 3).abs().max()) == 0 for l in range(3))


@pytest.mark.parametrize("t_on,t_all", [(0, 5), (7, 7), (3, 11), (130, 257), (601, 1102)])
def test_merged_split_launch_equals_the_single_split_passes_and_tracks_f32(t_on, t_all):
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(t_all * 5 + t_on)
    for name in ("off", "emo"):
        Ws, Bs = _net(g)
        _pack(L, eng, name, Ws, Bs)
    X = torch.randn(t_all * 104 * 32, generator=g).cuda()

    def bufs():
        f = lambda rows, dt=torch.float32: torch.full((max(t_all, 1) * rows * 32,), -3, dtype=dt, device="cuda")
        return dict(H=[f(192) for _ in range(3)], M=[torch.full((max(t_all, 1) * 3 * 64,), -3, dtype=torch.int32, device="cuda") for _ in range(3)],
                    z_off=f(4), z_emo=f(4))
    A, B, F = bufs(), bufs(), bufs()
    pa = _lib.ptr_array
    po, pe = _lib.ptr(eng.packed["off"]), _lib.ptr(eng.packed["emo"])
    so, se = _lib.ptr(eng.packed_split["off"]), _lib.ptr(eng.packed_split["emo"])
    # single split passes: off detached on [0, t_on) with colour rows 88, off saved on [t_on, t_all), emo saved on [0, t_on)
    _lib.check(L.esr_mlp_fwd_split(0, po, so, _lib.ptr(X), 0, t_on, pa(A["H"]), pa(A["M"]), 0, 88, _lib.ptr(A["z_off"]), s), "a")
    _lib.check(L.esr_mlp_fwd_split(0, po, so, _lib.ptr(X), t_on, t_all, pa(A["H"]), pa(A["M"]), 1, 0, _lib.ptr(A["z_off"]), s), "b")
    _lib.check(L.esr_mlp_fwd_split(0, pe, se, _lib.ptr(X), 0, t_on, pa(A["H"]), pa(A["M"]), 1, 0, _lib.ptr(A["z_emo"]), s), "c")
    _lib.check(L.esr_mlp_fwd_fine_split(po, so, pe, se, _lib.ptr(X), t_on, t_all, pa(B["H"]), pa(B["M"]), 88, _lib.ptr(B["z_off"]),
                                        _lib.ptr(B["z_emo"]), s), "merged")
    _lib.check(L.esr_mlp_fwd_fine(po, pe, _lib.ptr(X), t_on, t_all, pa(F["H"]), pa(F["M"]), 88, _lib.ptr(F["z_off"]),
                                  _lib.ptr(F["z_emo"]), s), "f32")
    torch.cuda.synchronize()
    for k in ("z_off", "z_emo"):
        assert torch.equal(A[k], B[k]), k                            # same kernel, same per-tile arithmetic: bit for bit
    for k in ("H", "M"):
        for l in range(3):
            assert torch.equal(A[k][l], B[k][l]), (k, l)
    n_on = t_on * 4 * 32
    assert rel_err(B["z_off"], F["z_off"]) < 3e-6
    if t_on:
        assert rel_err(B["z_emo"][:n_on], F["z_emo"][:n_on]) < 3e-6
    for l in range(3):
        assert rel_err(B["H"][l], F["H"][l]) < 3e-6, l


def test_trainer_step_with_split_forward_equals_the_f32_forward_step():
    """FineStep on a small slab scene: the split forward and the fp32 MFMA forward feed the same fp32 backward kernels;
    loss and all 23 gradients agree to 2e-5 of each gradient's largest value (a ReLU unit within rounding of zero may take
    the other branch: the scene's rays holding such a sample are not excluded here, hence not 1e-6)."""
    import numpy as np
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    from test_gpu_fine_path import build_gpu_model, gpu_batch
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=384, seed=9, mask="prune")
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    b = gpu_batch(sc)
    eng = m.engine
    assert eng.split_fwd
    loss_s, g_s = FineStep(m).forward_loss_backward(b, 60.0)
    torch.cuda.synchronize()
    loss_s, g_s = float(loss_s), {k: v.clone() for k, v in g_s.items()}
    eng.split_fwd = False
    loss_f, g_f = FineStep(m).forward_loss_backward(b, 60.0)
    torch.cuda.synchronize()
    eng.split_fwd = True
    assert abs(loss_s - float(loss_f)) < 2e-6 * max(1.0, abs(float(loss_f)))
    worst = max(rel_err(g_s[k], g_f[k]) for k in g_f)
    print("split vs f32 forward: worst gradient difference", worst)
    for k in g_f:
        assert rel_err(g_s[k], g_f[k]) < 5e-4, (k, rel_err(g_s[k], g_f[k]))
    assert np.median([rel_err(g_s[k], g_f[k]) for k in g_f]) < 2e-5


@pytest.mark.parametrize("tiles,gscale", [(1, 1.0), (37, 1e-4), (300, 1e-7), (64, 30.0)])
def test_split_dgrad_vs_double_precision_and_vs_the_f32_kernel(tiles, gscale):
    """The input-gradient chain from split planes (per-tile power-of-two scaling) against a double-precision evaluation with
    the SAME ReLU masks, and against mlp.hip's f32 kernel: dZ[0..2] and the grid-fed dX rows; output gradients of very different
    magnitudes per sample and per tile (weights of the compositor span orders of magnitude), rows above 43 left untouched."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(tiles + 11)
    Ws, Bs = _net(g)
    _pack(L, eng, "off", Ws, Bs)
    X = torch.randn(tiles, 104, 32, generator=g)
    Xd = X.cuda().contiguous()
    H = [torch.zeros(tiles, 192, 32, device="cuda") for _ in range(3)]
    M = [torch.zeros(tiles, 3, 64, dtype=torch.int32, device="cuda") for _ in range(3)]
    z = torch.zeros(tiles, 4, 32, device="cuda")
    _lib.check(L.esr_mlp_fwd(0, _lib.ptr(eng.packed["off"]), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H), _lib.ptr_array(M), 1, 0,
                             _lib.ptr(z), s), "fwd")
    # per-sample magnitudes over six decades, per-tile magnitudes over three
    dz = torch.randn(tiles, 4, 32, generator=g) * gscale
    dz *= 10.0 ** (-6.0 * torch.rand(tiles, 1, 32, generator=g)) * 10.0 ** (-3.0 * torch.rand(tiles, 1, 1, generator=g))
    dz[:, 3] = 0.0
    if tiles > 2:
        dz[1] = 0.0                                                   # an all-zero tile
    dzd = dz.cuda().contiguous()

    def run(split):
        dZ = [torch.full((tiles, 192, 32), -3.0, device="cuda") for _ in range(3)]
        dX = torch.full((tiles, 64, 32), 3.0, device="cuda")
        if split:
            amax = torch.zeros(1, device="cuda")
            rc = L.esr_mlp_dgrad_split(0, _lib.ptr(eng.packed_split["off"]), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(M),
                                       _lib.ptr_array(dZ), _lib.ptr(dX), _lib.ptr(amax), s)
            torch.cuda.synchronize()                                  # the launch's largest |dz| x the net's gain factor
            assert abs(float(amax) / amax_source(float(dz.abs().max()), Ws) - 1.0) < 1e-5
        else:
            rc = L.esr_mlp_dgrad(0, _lib.ptr(eng.packed["off"]), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(M), _lib.ptr_array(dZ),
                                 _lib.ptr(dX), s)
        _lib.check(rc, "dgrad")
        torch.cuda.synchronize()
        return dZ, dX
    dZs, dXs = run(True)
    dZf, dXf = run(False)
    # double-precision chain with the kernel's masks (h > 0 of the f32 forward's saved tiles)
    tm = lambda t, r: t.reshape(tiles, 32, r).permute(0, 2, 1).contiguous()
    rm = lambda t: t.permute(0, 2, 1).reshape(tiles * 32, t.shape[1])
    gcur = rm(dz[:, :3].double())
    masks = [(rm(H[l].cpu()) > 0).double() for l in range(3)]
    refs = []
    for l in (3, 2, 1):
        gcur = (gcur @ Ws[l].double()) * masks[l - 1]
        refs.append(gcur)
    dx_ref = gcur @ Ws[0].double()                                    # [n, 85] in the reference's input order
    rows = [r for r in range(44) if _in_colmap(0, r) >= 0]
    cols = [_in_colmap(0, r) for r in rows]
    for name, got_s, got_f, ref in (("dZ2", dZs[2], dZf[2], refs[0]), ("dZ1", dZs[1], dZf[1], refs[1]), ("dZ0", dZs[0], dZf[0], refs[2])):
        # per TILE: the error relative to the tile's largest gradient (what the per-tile scaling promises)
        r_t, s_t, f_t = tm(ref, 192), got_s.cpu().double(), got_f.cpu().double()
        scale = r_t.abs().amax(dim=(1, 2)).clamp_min(1e-300)
        es = float(((s_t - r_t).abs().amax(dim=(1, 2)) / scale).max())
        ef = float(((f_t - r_t).abs().amax(dim=(1, 2)) / scale).max())
        print(f"{name}: split {es:.2e}, f32 MFMA {ef:.2e} (per-tile max-norm)")
        assert es < 3e-6 and es < 4 * ef + 1e-6, (name, es, ef)
    dxr = tm(dx_ref[:, cols], len(rows))
    scale = dxr.abs().amax(dim=(1, 2)).clamp_min(1e-300)
    es = float(((dXs[:, rows].cpu().double() - dxr).abs().amax(dim=(1, 2)) / scale).max())
    ef = float(((dXf[:, rows].cpu().double() - dxr).abs().amax(dim=(1, 2)) / scale).max())
    print(f"dX: split {es:.2e}, f32 MFMA {ef:.2e}")
    assert es < 3e-6 and es < 4 * ef + 1e-6
    assert float((dXs[:, 44:] - 3.0).abs().max()) == 0.0                 # rows that lead to no grid: untouched
    if tiles > 2:
        assert float(dXs[1, :44].abs().max()) == 0.0 and float(dZs[0][1].abs().max()) == 0.0


@pytest.mark.parametrize("t_on,t_all", [(0, 5), (7, 7), (3, 11), (130, 257)])
def test_merged_split_dgrad_equals_the_single_passes(t_on, t_all):
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(t_all * 3 + t_on)
    nets = {}
    for name in ("off", "emo"):
        Ws, Bs = _net(g)
        nets[name] = Ws
        _pack(L, eng, name, Ws, Bs)
    M = [torch.randint(-2 ** 31, 2 ** 31 - 1, (t_all * 3 * 64,), generator=g, dtype=torch.int64).to(torch.int32).cuda() for _ in range(3)]
    dz = (torch.randn(t_all * 4 * 32, generator=g) * 1e-4).cuda()
    pa = _lib.ptr_array

    def bufs():
        f = lambda rows: torch.full((max(t_all, 1) * rows * 32,), -3.0, device="cuda")
        return dict(dZ=[f(192) for _ in range(3)], dX=f(64))
    A, B = bufs(), bufs()
    se, so = _lib.ptr(eng.packed_split["emo"]), _lib.ptr(eng.packed_split["off"])
    am = torch.zeros(2, device="cuda")
    _lib.check(L.esr_mlp_dgrad_split(0, se, _lib.ptr(dz), 0, t_on, pa(M), pa(A["dZ"]), _lib.ptr(A["dX"]), None, s), "emo")
    _lib.check(L.esr_mlp_dgrad_split(0, so, _lib.ptr(dz), t_on, t_all, pa(M), pa(A["dZ"]), _lib.ptr(A["dX"]), _lib.ptr(am[1:]), s), "off")
    _lib.check(L.esr_mlp_dgrad_fine_split(se, so, _lib.ptr(dz), t_on, t_all, pa(M), pa(B["dZ"]), _lib.ptr(B["dX"]), _lib.ptr(am), s), "merged")
    torch.cuda.synchronize()
    z3 = dz.view(t_all, 4, 32)[:, :3]                                   # (row 3 of the 4-row tile is padding: not part of the maximum)
    want_off = amax_source(float(z3[t_on:].abs().max()), nets["off"]) if t_all > t_on else 0.0
    want_emo = amax_source(float(z3[:t_on].abs().max()), nets["emo"]) if t_on else 0.0
    assert abs(float(am[0]) - max(want_off, want_emo)) <= 1e-5 * max(want_off, want_emo)
    assert abs(float(am[1]) - want_off) <= 1e-5 * want_off
    assert torch.equal(A["dX"], B["dX"])
    for l in range(3):
        assert torch.equal(A["dZ"][l], B["dZ"][l]), l


def _wgrad_operands(tiles, g, gscale, spread):
    """Random saved tiles of a 85-192-192-192-3 net's backward: H (>= 0, a third of them exactly zero like a ReLU's), dZ with
    per-sample magnitudes over `spread` decades times gscale, dz likewise; X with the stencil rows x5."""
    H = [torch.relu(torch.randn(tiles, 192, 32, generator=g) + 0.4) for _ in range(3)]
    mag = lambda: gscale * 10.0 ** (-spread * torch.rand(tiles, 1, 32, generator=g))
    dZ = [torch.randn(tiles, 192, 32, generator=g) * mag() * (1.0 + 3.0 * l) for l in range(3)]
    dz = torch.randn(tiles, 4, 32, generator=g) * mag()
    dz[:, 3] = 0.0
    X = torch.randn(tiles, 104, 32, generator=g)
    X[:, 7:31] *= 5.0
    return X, H, dZ, dz


@pytest.mark.parametrize("tiles,t0,crow,gscale,spread", [(1, 0, 0, 1.0, 0.0), (37, 5, 88, 1e-4, 4.0), (700, 0, 96, 1e-7, 6.0),
                                                         (1300, 11, 0, 3e-3, 3.0), (300, 0, 0, 50.0, 2.0)])
def test_split_wgrad_vs_double_precision_and_vs_the_f32_kernel(tiles, t0, crow, gscale, spread):
    """Weight gradients of the 192-wide net with the products on the 16-bit matrix cores (esr_wgrad_job_t::amax set;
    csrc/mlp.hip: wgrad_dma_body<..., SPLIT>) against float64 sums of the same fp32 operands, beside the f32 MFMA kernel on
    the same inputs: all four layers' dW and db, a tile range that does not start at 0, the three colour-row groups,
    gradient magnitudes from 1e-13 to 50.  Bar: max-norm error relative to the largest |dW| entry of the layer < 2e-6 and
    no more than 4x the f32 kernel's (+1e-7): the accuracy class of the kernel it replaces."""
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(tiles * 7 + crow)
    X, H, dZ, dz = _wgrad_operands(tiles, g, gscale, spread)
    rows = [r for r in range(96) if _in_colmap(0, r) >= 0]
    cols = [_in_colmap(0, r) for r in rows]
    src_rows = [r + crow if r < 6 else r for r in rows]
    rm = lambda t: t[t0:].permute(0, 2, 1).reshape((tiles - t0) * 32, t.shape[1]).double()
    x_ref = torch.zeros((tiles - t0) * 32, 85, dtype=torch.float64)
    x_ref[:, cols] = rm(X[:, src_rows])
    A = [rm(dZ[0]), rm(dZ[1]), rm(dZ[2]), rm(dz[:, :3])]
    Bm = [x_ref, rm(H[0]), rm(H[1]), rm(H[2])]
    want_w = [a.t() @ b for a, b in zip(A, Bm)]
    want_b = [a.sum(0) for a in A]
    dev = lambda t: t.cuda().contiguous()
    Xd, Hd, dZd, dzd = dev(X), [dev(h) for h in H], [dev(z) for z in dZ], dev(dz)
    scratch = torch.empty(L.esr_mlp_wgrad_scratch_floats(), device="cuda")
    amax = torch.zeros(1, device="cuda")
    _lib.check(L.esr_absmax(_lib.ptr(dzd[t0:]), C.c_int64((tiles - t0) * 128), _lib.ptr(amax), s), "absmax")
    assert float(amax) == float(dz[t0:].abs().max())

    def run(split):
        gw = [torch.zeros(sh, device="cuda") for sh in ((192, 85), (192, 192), (192, 192), (3, 192))]
        gb = [torch.zeros(n, device="cuda") for n in (192, 192, 192, 3)]
        jobs = (_lib.EsrWgradJob * 1)()
        ptrs = [_lib.ptr_array(Hd), _lib.ptr_array(dZd), _lib.ptr_array(gw), _lib.ptr_array(gb)]
        jb = jobs[0]
        jb.kind, jb.color_row0, jb.t0, jb.t1 = 0, crow, t0, tiles
        jb.X, jb.dz = Xd.data_ptr(), dzd.data_ptr()
        jb.H, jb.dZ, jb.gw, jb.gb = (C.addressof(p) for p in ptrs)
        if split:
            jb.amax = amax.data_ptr()
        _lib.check(L.esr_mlp_wgrad_batch(jobs, 1, 0, _lib.ptr(scratch), C.c_int64(scratch.numel()), s), "wgrad")
        torch.cuda.synchronize()
        return [w.cpu().double() for w in gw], [b.cpu().double() for b in gb]
    ws, bs = run(True)
    wf, bf_ = run(False)
    for l in range(4):
        scale = float(want_w[l].abs().max())
        es, ef = float((ws[l] - want_w[l]).abs().max()) / scale, float((wf[l] - want_w[l]).abs().max()) / scale
        print(f"layer {l}: dW split {es:.2e}, f32 MFMA {ef:.2e} of the largest entry ({scale:.2e})")
        assert es < 2e-6 and es < 4 * ef + 1e-7, (l, es, ef)
        sb = float(want_b[l].abs().max()) + 1e-300
        assert float((bs[l] - want_b[l]).abs().max()) / sb < 1e-5


@pytest.mark.parametrize("kind,tiles,crow,save", [(1, 1, 0, 2), (1, 37, 0, 1), (1, 700, 0, 2), (2, 1, 0, 1), (2, 41, 88, 1), (2, 600, 96, 1),
                                                  (3, 5, 0, 1), (3, 333, 88, 0), (0, 9, 0, 1)])
def test_split_kernels_for_every_net_vs_double_precision_and_vs_the_f32_kernels(kind, tiles, crow, save):
    """esr_mlp_fwd_split / esr_mlp_dgrad_split for the tone mapper (33-192-3, two layers), the BRDF net (76-128-128-128-5, an
    8-row output tile) and the emission net (76-128-128-128-3) -- the 128-wide nets run 7 steps per tile group, i.e. every
    other group starts in the second LDS buffer, and finish a layer's last tile inside the 18 slots before the next layer
    reads it -- against a float64 chain and beside the f32 MFMA kernels on the same buffers: outputs, saved tiles, masks
    (equal except where a pre-activation is within rounding of zero), hidden and input gradients."""
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr("cuda:0")
    n = NET[kind]
    in_dim, xrows, nl, hid, nout, zrows = n["in_dim"], n["xrows"], n["nl"], n["hid"], n["out"], n["zrows"]
    g = torch.Generator().manual_seed(kind * 1000 + tiles)
    dims = [in_dim] + [hid] * (nl - 1) + [nout]
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(nl)]
    Bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(nl)]
    X = torch.randn(tiles, xrows, 32, generator=g)
    rows = [r for r in range(min(xrows, 96)) if _in_colmap(kind, r) >= 0]
    cols = [_in_colmap(kind, r) for r in rows]
    cw = n.get("cw", 6)
    src_rows = [r + crow if r < cw else r for r in rows]
    x_ref = torch.zeros(tiles * 32, in_dim, dtype=torch.float64)
    x_ref[:, cols] = X[:, src_rows, :].permute(0, 2, 1).reshape(tiles * 32, len(rows)).double()
    h, hs, pres = x_ref, [], []
    for i in range(nl):
        h = torch.nn.functional.linear(h, Ws[i].double(), Bs[i].double())
        if i + 1 < nl:
            pres.append(h)
            h = torch.relu(h)
            hs.append(h)
    tm = lambda t, r: t.reshape(tiles, 32, r).permute(0, 2, 1).contiguous()
    # pack: fp32 buffer + split planes in one batch launch
    keep = [(a.cuda().contiguous(), b.cuda().contiguous()) for a, b in zip(Ws, Bs)]
    w = _lib.EsrMlpWeights()
    for i, (a, b) in enumerate(keep):
        w.w[i], w.b[i] = a.data_ptr(), b.data_ptr()
    packed = torch.empty(L.esr_mlp_packed_floats(kind), device="cuda")
    planes = torch.empty(L.esr_mlp_packed_split_elems(kind), dtype=torch.float16, device="cuda")
    kinds = (C.c_int32 * 1)(kind)
    wsp = (C.c_void_p * 1)(C.addressof(w))
    p32 = (C.c_void_p * 1)(packed.data_ptr())
    psp = (C.c_void_p * 1)(planes.data_ptr())
    _lib.check(L.esr_mlp_pack_batch(1, kinds, wsp, p32, None, psp, s), "pack")
    Xd = X.cuda().contiguous()

    def fwd(split):
        H = [torch.full((tiles, hid, 32), -3.0, device="cuda") for _ in range(nl - 1)]
        M = [torch.full((tiles, hid // 64, 64), -3, dtype=torch.int32, device="cuda") for _ in range(nl - 1)]
        z = torch.full((tiles, zrows, 32), 7.0, device="cuda")
        if split:
            rc = L.esr_mlp_fwd_split(kind, _lib.ptr(packed), _lib.ptr(planes), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H),
                                     _lib.ptr_array(M), save, crow, _lib.ptr(z), s)
        else:
            rc = L.esr_mlp_fwd(kind, _lib.ptr(packed), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H), _lib.ptr_array(M), save, crow,
                               _lib.ptr(z), s)
        _lib.check(rc, "fwd")
        torch.cuda.synchronize()
        return H, M, z
    Hs, Ms, zs = fwd(True)
    Hf, Mf, zf = fwd(False)
    zr = tm(h, nout)
    scale = float(zr.abs().max())
    es, ef = float((zs[:, :nout].cpu().double() - zr).abs().max()) / scale, float((zf[:, :nout].cpu().double() - zr).abs().max()) / scale
    print(f"kind {kind}: z split {es:.2e}, f32 MFMA {ef:.2e}")
    assert es < 3e-6 and es < 4 * ef + 1e-6
    assert float(zs[:, nout:].abs().max()) == 0.0                       # padding rows of the output tile: zeros
    if save == 1:
        for l in range(nl - 1):
            hr = tm(hs[l], hid)
            sc = float(hr.abs().max())
            assert float((Hs[l].cpu().double() - hr).abs().max()) / sc < 3e-6, l
    else:
        for l in range(nl - 1):
            assert float((Hs[l] + 3.0).abs().max()) == 0.0              # not saved: untouched
    if save:
        for l in range(nl - 1):
            # bit (it & 1) * 16 + r of word [tile][it >> 1][lane]: compare the two kernels bit for bit away from the knife edge
            edge = tm((pres[l].abs() < 1e-5).double(), hid).sum(dim=(1, 2)) > 0
            same = (Ms[l] == Mf[l]).all(dim=(1, 2)).cpu()
            assert bool((same | edge).all()), l
    # input gradients from the split forward's masks, both kernels
    dz = torch.randn(tiles, zrows, 32, generator=g) * 1e-3 * 10.0 ** (-3.0 * torch.rand(tiles, 1, 32, generator=g))
    dz[:, nout:] = 0.0
    dzd = dz.cuda().contiguous()
    if not save:
        return

    def bwd(split):
        dZ = [torch.full((tiles, hid, 32), -3.0, device="cuda") for _ in range(nl - 1)]
        dX = torch.full((tiles, 64, 32), 3.0, device="cuda")
        if split:
            rc = L.esr_mlp_dgrad_split(kind, _lib.ptr(planes), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Ms), _lib.ptr_array(dZ),
                                       _lib.ptr(dX), None, s)
        else:
            rc = L.esr_mlp_dgrad(kind, _lib.ptr(packed), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Ms), _lib.ptr_array(dZ),
                                 _lib.ptr(dX), s)
        _lib.check(rc, "dgrad")
        torch.cuda.synchronize()
        return dZ, dX
    dZs, dXs = bwd(True)
    dZf, dXf = bwd(False)
    n_dx = {0: 44, 1: 36, 2: 44, 3: 44}[kind]
    for name, a, b in [(f"dZ{l}", dZs[l], dZf[l]) for l in range(nl - 1)] + [("dX", dXs[:, :n_dx], dXf[:, :n_dx])]:
        a_, b_ = a.cpu().double(), b.cpu().double()
        scale_t = b_.abs().amax(dim=(1, 2)).clamp_min(1e-300)            # per tile: what the per-tile scaling promises
        e = float(((a_ - b_).abs().amax(dim=(1, 2)) / scale_t).max())
        print(f"kind {kind}: {name} split vs f32 MFMA {e:.2e} (per-tile max-norm)")
        assert e < 4e-6, (name, e)
    assert float((dXs[:, n_dx:] - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("kind,tiles,t0,crow,gscale", [(2, 1, 0, 0, 1.0), (2, 500, 3, 88, 1e-5), (3, 900, 0, 96, 1e-3), (1, 300, 0, 0, 1e-4)])
def test_split_wgrad_of_the_other_nets_vs_double_precision(kind, tiles, t0, crow, gscale):
    """esr_wgrad_job_t::amax on the 128-wide nets (BRDF: 5 outputs in an 8-row tile; emission) and the tone mapper: their
    layer shapes run the per-shape LDS-DMA kernels with the SPLIT products (csrc/mlp.hip: launch_cfg<2>) -- against float64
    sums, beside the f32 MFMA kernels."""
    from esr_nerf_amd import _lib
    L = _lib.lib()
    s = _lib.stream_ptr("cuda:0")
    n = NET[kind]
    in_dim, xrows, nl, hid, nout, zrows = n["in_dim"], n["xrows"], n["nl"], n["hid"], n["out"], n["zrows"]
    g = torch.Generator().manual_seed(kind * 77 + tiles)
    H = [torch.relu(torch.randn(tiles, hid, 32, generator=g) + 0.4) for _ in range(nl - 1)]
    mag = lambda: gscale * 10.0 ** (-3.0 * torch.rand(tiles, 1, 32, generator=g))
    dZ = [torch.randn(tiles, hid, 32, generator=g) * mag() * (1.0 + 3.0 * l) for l in range(nl - 1)]
    dz = torch.randn(tiles, zrows, 32, generator=g) * mag()
    dz[:, nout:] = 0.0
    X = torch.randn(tiles, xrows, 32, generator=g)
    rows = [r for r in range(min(xrows, 96)) if _in_colmap(kind, r) >= 0]
    cols = [_in_colmap(kind, r) for r in rows]
    src_rows = [r + crow if r < 6 else r for r in rows]
    rm = lambda t: t[t0:].permute(0, 2, 1).reshape((tiles - t0) * 32, t.shape[1]).double()
    x_ref = torch.zeros((tiles - t0) * 32, in_dim, dtype=torch.float64)
    x_ref[:, cols] = rm(X[:, src_rows])
    A = [rm(z) for z in dZ] + [rm(dz[:, :nout])]
    Bm = [x_ref] + [rm(h) for h in H]
    want_w = [a.t() @ b for a, b in zip(A, Bm)]
    dev = lambda t: t.cuda().contiguous()
    Xd, Hd, dZd, dzd = dev(X), [dev(h) for h in H], [dev(z) for z in dZ], dev(dz)
    scratch = torch.empty(L.esr_mlp_wgrad_scratch_floats(), device="cuda")
    amax = torch.zeros(1, device="cuda")
    _lib.check(L.esr_absmax(_lib.ptr(dzd), C.c_int64(dzd.numel()), _lib.ptr(amax), s), "absmax")
    dims = [in_dim] + [hid] * (nl - 1) + [nout]

    def run(split):
        gw = [torch.zeros(dims[i + 1], dims[i], device="cuda") for i in range(nl)]
        gb = [torch.zeros(dims[i + 1], device="cuda") for i in range(nl)]
        jobs = (_lib.EsrWgradJob * 1)()
        ptrs = [_lib.ptr_array(Hd), _lib.ptr_array(dZd), _lib.ptr_array(gw), _lib.ptr_array(gb)]
        jb = jobs[0]
        jb.kind, jb.color_row0, jb.t0, jb.t1 = kind, crow, t0, tiles
        jb.X, jb.dz = Xd.data_ptr(), dzd.data_ptr()
        jb.H, jb.dZ, jb.gw, jb.gb = (C.addressof(p) for p in ptrs)
        if split:
            jb.amax = amax.data_ptr()
        _lib.check(L.esr_mlp_wgrad_batch(jobs, 1, 0, _lib.ptr(scratch), C.c_int64(scratch.numel()), s), "wgrad")
        torch.cuda.synchronize()
        return [w.cpu().double() for w in gw], [b.cpu().double() for b in gb]
    ws, bs = run(True)
    wf, _ = run(False)
    for l in range(nl):
        scale = float(want_w[l].abs().max())
        es, ef = float((ws[l] - want_w[l]).abs().max()) / scale, float((wf[l] - want_w[l]).abs().max()) / scale
        print(f"kind {kind} layer {l}: dW split {es:.2e}, f32 MFMA {ef:.2e}")
        assert es < 2e-6 and es < 4 * ef + 1e-7, (l, es, ef)
        assert float((bs[l] - A[l].sum(0)).abs().max()) / (float(A[l].sum(0).abs().max()) + 1e-300) < 1e-5


def _fwd_once(eng, L, s, g, tiles, scale=1.0, poke=None):
    X = torch.randn(tiles, 104, 32, generator=g) * scale
    if poke is not None:
        X[poke[0], poke[1], poke[2]] = poke[3]
    X = X.cuda().contiguous()
    H = [torch.zeros(tiles, 192, 32, device="cuda") for _ in range(3)]
    M = [torch.zeros(tiles, 3, 64, dtype=torch.int32, device="cuda") for _ in range(3)]
    z = torch.zeros(tiles, 4, 32, device="cuda")
    _lib_ = __import__("esr_nerf_amd._lib", fromlist=["_lib"])
    _lib_.check(L.esr_mlp_fwd_split(0, _lib_.ptr(eng.packed["off"]), _lib_.ptr(eng.packed_split["off"]), _lib_.ptr(X), 0, tiles,
                                    _lib_.ptr_array(H), _lib_.ptr_array(M), 1, 0, _lib_.ptr(z), s), "fwd")
    torch.cuda.synchronize()
    return H


def test_range_flag_is_raised_by_hidden_activations_inputs_weights_and_gains_and_by_nothing_else():
    """What the split kernels' first planes cannot carry raises the device's sticky range flag (esr_mlp_split_range_flag):
    a hidden activation >= 60000 (inputs of 3e4 against weights of ~0.1), ONE input value beyond the range (whose products a
    zero weight column would hide from the activations), a weight with |64 w| beyond fp16 (esr_mlp_pack_batch), and a net
    whose gradient gain bound exceeds 2^18 (split_gain_kernel).  In-range data leaves it alone, and the flag reaches the
    engine through range_probe / range_hit, which clears it."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    assert eng.split_fwd and eng.range_flag is not None
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(5)
    Ws, Bs = _net(g)
    _pack(L, eng, "off", Ws, Bs)
    eng.range_flag.zero_()
    _fwd_once(eng, L, s, g, 9)
    assert int(eng.range_flag) == 0
    H = _fwd_once(eng, L, s, g, 9, scale=3.0e4)
    assert float(H[0].max()) > 6.0e4                                   # (the fp32 epilogue still shows the magnitude)
    assert int(eng.range_flag) == 1
    eng.range_probe()
    assert eng.range_hit() and int(eng.range_flag) == 0 and eng.split_fallback_steps == 1
    eng.range_probe()
    assert not eng.range_hit()
    # one input value beyond the range, in a row whose weights are all zero: no hidden activation shows it
    row = next(r for r in range(6, 96) if _in_colmap(0, r) >= 0)
    Wz = [w.clone() for w in Ws]
    Wz[0][:, _in_colmap(0, row)] = 0.0
    _pack(L, eng, "off", Wz, Bs)
    H = _fwd_once(eng, L, s, g, 5, poke=(3, row, 17, -7.0e4))
    assert float(H[0].max()) < 100.0 and int(eng.range_flag) == 1
    eng.range_flag.zero_()
    # a weight beyond 1023: its first plane fp16(64 w) is inf -- raised by the packing launch itself, for any layer (also the
    # output layer, whose results never become planes)
    for layer, val in ((3, 1100.0), (0, -2000.0), (2, float("inf"))):
        Wb = [w.clone() for w in Ws]
        Wb[layer][1, 5] = val
        _pack(L, eng, "off", Wb, Bs)
        torch.cuda.synchronize()
        assert int(eng.range_flag) == 1, layer
        eng.range_flag.zero_()
    Wb = [w.clone() for w in Ws]
    Wb[3][1, 5] = 1000.0                                               # 64000 is an fp16 number
    _pack(L, eng, "off", Wb, Bs)
    torch.cuda.synchronize()
    assert int(eng.range_flag) == 0
    # a net whose weights are in range one by one but whose gain bound is beyond 2^18
    Wg = [w * 20.0 for w in Ws]
    assert gain_bound(Wg) > 262144.0 and max(float(w.abs().max()) for w in Wg) < 1000.0
    _pack(L, eng, "off", Wg, Bs)
    torch.cuda.synchronize()
    assert int(eng.range_flag) == 1
    eng.range_flag.zero_()
    _pack(L, eng, "off", Ws, Bs)
    torch.cuda.synchronize()
    assert int(eng.range_flag) == 0
    # the bound itself, behind the planes
    go = int(L.esr_mlp_split_gain_offset(0))
    got = float(eng.packed_split["off"][go: go + 2].view(torch.float32)[0])
    assert abs(got / gain_bound(Ws) - 1.0) < 1e-5


@pytest.mark.parametrize("wscale,gscale", [(1.0, 1.0), (4.0, 1e-3), (9.0, 1e-6), (0.05, 20.0)])
def test_split_backward_cannot_overflow_whatever_the_weights_gain(wscale, gscale):
    """The input-gradient chain and the weight gradients that follow it run scaled by the net's GAIN BOUND (split_gain_kernel):
    with weights 4x / 9x their initial size a hidden gradient exceeds the output gradient by 1e3 .. 1e5 -- far beyond the fixed
    headroom the scales used to leave -- and every value must still be finite and as accurate (relative to the tile's / the
    layer's largest entry) as the f32 MFMA kernels'.  Masks all ones (the worst case: nothing is cut)."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(int(wscale * 10))
    Ws, Bs = _net(g)
    Ws = [w.abs() * wscale for w in Ws]                                # same-sign weights: the bound is nearly attained
    eng.range_flag.zero_()
    _pack(L, eng, "off", Ws, Bs)
    tiles = 70
    M = [torch.full((tiles, 3, 64), -1, dtype=torch.int32, device="cuda") for _ in range(3)]
    dz = torch.rand(tiles, 4, 32, generator=g) * gscale
    dz[:, 3] = 0.0
    dzd = dz.cuda().contiguous()
    dZ = [torch.zeros(tiles, 192, 32, device="cuda") for _ in range(3)]
    dX = torch.zeros(tiles, 64, 32, device="cuda")
    amax = torch.zeros(1, device="cuda")
    _lib.check(L.esr_mlp_dgrad_split(0, _lib.ptr(eng.packed_split["off"]), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(M),
                                     _lib.ptr_array(dZ), _lib.ptr(dX), _lib.ptr(amax), s), "dgrad")
    torch.cuda.synchronize()
    assert int(eng.range_flag) == 0                                    # (9x: the gain bound is still below 2^18)
    rm = lambda t: t.permute(0, 2, 1).reshape(tiles * 32, t.shape[1])
    tm = lambda t, r: t.reshape(tiles, 32, r).permute(0, 2, 1).contiguous()
    gcur = rm(dz[:, :3].double())
    growth = 0.0
    for l, got in ((3, dZ[2]), (2, dZ[1]), (1, dZ[0])):
        gcur = gcur @ Ws[l].double()
        ref = tm(gcur, 192)
        assert bool(torch.isfinite(got).all())
        scale = ref.abs().amax(dim=(1, 2)).clamp_min(1e-300)
        e = float(((got.cpu().double() - ref).abs().amax(dim=(1, 2)) / scale).max())
        growth = max(growth, float(ref.abs().max()) / float(dz.abs().max()))
        print(f"wscale {wscale}: dZ{l - 1} error {e:.2e} of the tile's largest; growth over dz {growth:.1f}")
        assert e < 3e-6, (l, e)
    assert growth <= gain_bound(Ws) * (1 + 1e-6)
    # the scale source the weight gradients get covers every hidden gradient with their smallest headroom (32x)
    assert float(amax) * 32.0 >= float(max(z.abs().max() for z in dZ))
    # ... and the split weight-gradient launch on these operands is finite and accurate
    X = torch.randn(tiles, 104, 32, generator=g)
    H = [torch.relu(torch.randn(tiles, 192, 32, generator=g) + 0.4) for _ in range(3)]
    Xd, Hd = X.cuda().contiguous(), [h.cuda().contiguous() for h in H]
    gw = [torch.zeros(sh, device="cuda") for sh in ((192, 85), (192, 192), (192, 192), (3, 192))]
    gb = [torch.zeros(n, device="cuda") for n in (192, 192, 192, 3)]
    scratch = torch.empty(L.esr_mlp_wgrad_scratch_floats(), device="cuda")
    jobs = (_lib.EsrWgradJob * 1)()
    ptrs = [_lib.ptr_array(Hd), _lib.ptr_array(dZ), _lib.ptr_array(gw), _lib.ptr_array(gb)]
    jb = jobs[0]
    jb.kind, jb.color_row0, jb.t0, jb.t1 = 0, 0, 0, tiles
    jb.X, jb.dz = Xd.data_ptr(), dzd.data_ptr()
    jb.H, jb.dZ, jb.gw, jb.gb = (C.addressof(p) for p in ptrs)
    jb.amax = amax.data_ptr()
    _lib.check(L.esr_mlp_wgrad_batch(jobs, 1, 0, _lib.ptr(scratch), C.c_int64(scratch.numel()), s), "wgrad")
    torch.cuda.synchronize()
    for l in (1, 2):
        want = rm(dZ[l].cpu().double()).t() @ rm(H[l - 1].double())
        e = float((gw[l].cpu().double() - want).abs().max()) / float(want.abs().max())
        print(f"wscale {wscale}: dW{l} error {e:.2e}")
        assert bool(torch.isfinite(gw[l]).all()) and e < 2e-6, (l, e)


def _heal_case(make_bad, steps_before=1):
    """A FineStep run whose step `steps_before` overflows the split kernels' range: the step object must hand back the SAME
    loss and gradients as an engine that runs every product on the f32 MFMA kernels from the start, without an exception,
    and count one fallback; the steps before and after it run on the split kernels."""
    import warnings
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    from test_gpu_fine_path import build_gpu_model, gpu_batch
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=384, seed=9, mask="prune")
    b = gpu_batch(sc)
    out = {}
    for mode in ("split", "f32"):
        m = build_gpu_model(sc, seed=1, grid_seed=2)
        eng = m.engine
        if mode == "f32":
            eng.split_fwd = eng.split_bwd = eng.split_wgrad = eng.split_tone_wgrad = False
        else:
            assert eng.split_fwd
            eng.range_flag.zero_()
        step = FineStep(m)
        for _ in range(steps_before):
            step.forward_loss_backward(b, 60.0)
        assert eng.split_fallback_steps == 0
        undo = make_bad(m)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            loss, grads = step.forward_loss_backward(b, 60.0)
        torch.cuda.synchronize()
        out[mode] = (float(loss), {k: v.clone() for k, v in grads.items()})
        if mode == "split":
            assert eng.split_fallback_steps == 1 and int(eng.range_flag) == 0
            assert eng.split_fwd and eng.split_bwd and eng.split_wgrad          # the fallback is per step
            undo()
            step.forward_loss_backward(b, 60.0)                         # back in range: back on the split kernels
            torch.cuda.synchronize()
            assert eng.split_fallback_steps == 1
        step.close()
    (ls, gs), (lf, gf) = out["split"], out["f32"]
    assert ls == lf or abs(ls - lf) <= 1e-6 * abs(lf), (ls, lf)
    for k in gf:
        assert bool(torch.isfinite(gs[k]).all()), k
        # the same f32 kernels on the same data; only the order of the float atomics differs between two runs
        assert rel_err(gs[k], gf[k]) < 2e-5, (k, rel_err(gs[k], gf[k]))
    return gs


def test_trainer_step_heals_a_hidden_activation_overflow_in_the_same_step():
    """A first-layer bias of 7e4 in the emo net pushes its hidden activations beyond fp16's range in the middle of a run: the
    step that sees it is re-run on the f32 MFMA kernels before its gradients leave the step object."""
    def make_bad(m):
        lin = m.emo_rgbnet.layers()[0]
        keep = lin.bias.data.clone()
        lin.bias.data[3] = 7.0e4

        def undo():
            lin.bias.data.copy_(keep)
        return undo
    _heal_case(make_bad)


def test_trainer_step_heals_a_weight_beyond_the_planes_range():
    """One output-layer weight of 1500 (64 w is not an fp16 number): raised by the packing launch of the step, healed in it."""
    def make_bad(m):
        lin = m.off_rgbnet.layers()[3]
        keep = lin.weight.data.clone()
        lin.weight.data[1, 7] = 1500.0

        def undo():
            lin.weight.data.copy_(keep)
        return undo
    _heal_case(make_bad, steps_before=2)


def test_strict_mode_raises_instead_of_falling_back():
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    from test_gpu_fine_path import build_gpu_model, gpu_batch
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=128, seed=3)
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    eng = m.engine
    eng.range_flag.zero_()
    eng.split_strict = True
    m.emo_rgbnet.layers()[0].bias.data[3] = 7.0e4
    with pytest.raises(RuntimeError, match="fp16's range"):
        FineStep(m).forward_loss_backward(gpu_batch(sc), 60.0)
    torch.cuda.synchronize()
    assert int(eng.range_flag) == 0


def test_a_persistent_cause_switches_the_engine_to_the_f32_kernels_and_strict_mode_defers_under_data_parallelism():
    """(i) a cause that persists (a weight of 1500) would make EVERY step run the split attempt, wait, and run again on
    the f32 MFMA kernels; after three consecutive fallbacks the engine switches to those kernels for good and warns a second
    time -- the following steps run once.  (ii) ESR_SPLIT_STRICT=1 under data parallelism (``defer_overflow``): raising inside
    one rank's step would leave the others waiting in the gradient exchange; the hit is carried by the overflow word
    instead (every rank raises together at its next check)."""
    import warnings
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    from test_gpu_fine_path import build_gpu_model, gpu_batch
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=128, seed=3)
    b = gpu_batch(sc)
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    eng = m.engine
    eng.range_flag.zero_()
    m.off_rgbnet.layers()[3].weight.data[1, 7] = 1500.0
    step = FineStep(m)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always", RuntimeWarning)
        for i in range(5):
            calls0 = eng.n_calls
            loss, grads = step.forward_loss_backward(b, 60.0)
            torch.cuda.synchronize()
            assert all(bool(torch.isfinite(v).all()) for v in grads.values())
            if i == 0:
                twice = eng.n_calls - calls0
            if i >= 3:                                                     # one pass over the step's launches, not two
                assert eng.n_calls - calls0 < 0.7 * twice, (i, eng.n_calls - calls0, twice)
    assert eng.split_fallback_steps == 3 and not (eng.split_fwd or eng.split_bwd or eng.split_wgrad or eng.split_tone_wgrad)
    msgs = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning)]
    assert len(msgs) == 2 and "now runs every MLP launch on the f32 MFMA kernels" in msgs[1], msgs
    # (ii)
    m2 = build_gpu_model(sc, seed=1, grid_seed=2)
    e2 = m2.engine
    e2.range_flag.zero_()
    e2.split_strict, e2.defer_overflow = True, True
    m2.emo_rgbnet.layers()[0].bias.data[3] = 7.0e4
    FineStep(m2).forward_loss_backward(b, 60.0)                            # no exception on this rank ...
    torch.cuda.synchronize()
    assert e2.overflow_seen and getattr(e2, "range_strict_seen", False)     # ... the hit rides on the overflow word


def test_autograd_route_heals_in_the_forward():
    """VoxurfF.forward (the drop-in route: results go to the caller's torch code) waits for the range probe at the end of the
    forward and re-runs it on the f32 MFMA kernels; the backward of that call follows on the f32 kernels."""
    import warnings
    from esr_nerf_amd.synthetic import slab_scene
    from test_gpu_fine_path import build_gpu_model, gpu_batch
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=256, seed=4)
    b = gpu_batch(sc)
    res = {}
    for mode in ("split", "f32"):
        m = build_gpu_model(sc, seed=1, grid_seed=2)
        m.train()
        eng = m.engine
        if mode == "f32":
            eng.split_fwd = eng.split_bwd = eng.split_wgrad = eng.split_tone_wgrad = False
        else:
            eng.range_flag.zero_()
        m.off_rgbnet.layers()[1].weight.data[5, 9] = -3000.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out = m(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=60.0)
        loss = (out["srgb/rgb"] ** 2).mean() + (out["lin/rgb"] ** 2).mean() + out["etc/alphainv_cum"].mean()
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        if mode == "split":
            assert eng.split_fallback_steps == 1 and eng.split_fwd
    assert abs(res["split"][0] - res["f32"][0]) <= 1e-6 * abs(res["f32"][0])
    for k, v in res["f32"][1].items():
        assert bool(torch.isfinite(res["split"][1][k]).all()), k
        assert rel_err(res["split"][1][k], v) < 2e-5, k




@pytest.mark.parametrize("engine", ["split", "f32"])
def test_tone_wgrad_takes_the_forwards_branches(engine):
    """The tone mapper's weight gradients RECOMPUTE the hidden layer (csrc/tone_wgrad.hip) instead of reading what the forward
    saved.  The step is self-consistent only if the recomputation takes, at every (sample, unit), the ReLU branch the forward
    took -- also at a unit within summation noise of its kink, where any other order of the same sum may land on the other
    side.  Round 6 made the recomputation the forward's arithmetic bit for bit (same products, same order, same fma, same integer
    ReLU).  Adversarial data: every sample is one of eight vectors and every unit's bias cancels its pre-activation on one of
    them, so an eighth of ALL (sample, unit) pairs sit within rounding of zero; the reference is a float64 evaluation ON THE
    BRANCHES THE FORWARD SAVED in its masks.  One disagreeing pair would show as that sample's whole contribution to the
    unit's row (|dz W1| |x| ~ 1e-3 of the row's largest entry, against the 2e-6 asserted)."""
    from esr_nerf_amd import _lib
    from decisions import _decode_masks
    L = _lib.lib()
    s = _lib.stream_ptr("cuda:0")
    tiles, kind = 96, 1
    g = torch.Generator().manual_seed(5)
    W0 = torch.randn(192, 33, generator=g) / 33 ** 0.5
    W1 = torch.randn(3, 192, generator=g) / 192 ** 0.5
    b1 = torch.randn(3, generator=g) * 0.1
    proto = torch.randn(8, 33, generator=g)
    which = torch.randint(0, 8, (tiles * 32,), generator=g)
    x = proto[which]                                                    # [samples, 33]
    # b0[u] = -fp32(W0[u] . proto[u % 8]): summed in fp32 in yet another order than either kernel
    b0 = -(W0 * proto[torch.arange(192) % 8]).sum(1)
    X = torch.zeros(tiles, 48, 32)
    X[:, :33] = x.reshape(tiles, 32, 33).permute(0, 2, 1)
    pre = torch.nn.functional.linear(x.double(), W0.double(), b0.double())        # [samples, 192]
    on_edge = pre.abs() < 1e-6
    assert int(on_edge.sum()) > tiles * 32 * 192 // 10
    keep = [(W0.cuda().contiguous(), b0.cuda().contiguous()), (W1.cuda().contiguous(), b1.cuda().contiguous())]
    w = _lib.EsrMlpWeights()
    for i, (a, b) in enumerate(keep):
        w.w[i], w.b[i] = a.data_ptr(), b.data_ptr()
    packed = torch.empty(L.esr_mlp_packed_floats(kind), device="cuda")
    planes = torch.empty(L.esr_mlp_packed_split_elems(kind), dtype=torch.float16, device="cuda")
    _lib.check(L.esr_mlp_pack_batch(1, (C.c_int32 * 1)(kind), (C.c_void_p * 1)(C.addressof(w)), (C.c_void_p * 1)(packed.data_ptr()), None,
                                    (C.c_void_p * 1)(planes.data_ptr()), s), "pack")
    Xd = X.cuda().contiguous()
    M = [torch.zeros(tiles, 3, 64, dtype=torch.int32, device="cuda")]
    H = [torch.zeros(tiles, 192, 32, device="cuda")]
    z = torch.zeros(tiles, 4, 32, device="cuda")
    if engine == "split":
        rc = L.esr_mlp_fwd_split(kind, _lib.ptr(packed), _lib.ptr(planes), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H), _lib.ptr_array(M), 2, 0,
                                 _lib.ptr(z), s)
    else:
        rc = L.esr_mlp_fwd(kind, _lib.ptr(packed), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(H), _lib.ptr_array(M), 2, 0, _lib.ptr(z), s)
    _lib.check(rc, "fwd")
    torch.cuda.synchronize()
    mask = _decode_masks(M[0].cpu(), 6).permute(0, 2, 1).reshape(tiles * 32, 192)              # the forward's branches
    frac_on = float(mask[on_edge].double().mean())
    print(f"{engine}: {int(on_edge.sum())} (sample, unit) pairs within 1e-6 of the kink, the forward kept {frac_on:.2f} of them")
    assert 0.1 < frac_on < 0.9                                          # (noise decides there: the test has teeth)
    dz = torch.randn(tiles, 4, 32, generator=g) * 1e-3
    dz[:, 3] = 0.0
    dzd = dz.cuda().contiguous()
    gw0, gb0 = torch.zeros(192, 33, device="cuda"), torch.zeros(192, device="cuda")
    gw1, gb1 = torch.zeros(3, 192, device="cuda"), torch.zeros(3, device="cuda")
    scratch = torch.empty(L.esr_tone_wgrad_scratch_floats(), device="cuda")
    amax = dzd.abs().max().reshape(1).contiguous()
    if engine == "split":
        rc = L.esr_tone_wgrad_recompute_split(_lib.ptr(Xd), _lib.ptr(dzd), _lib.ptr(keep[0][0]), _lib.ptr(keep[0][1]), _lib.ptr(keep[1][0]),
                                              _lib.ptr(amax), 0, tiles, _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gw1), _lib.ptr(gb1),
                                              _lib.ptr(scratch), C.c_int64(scratch.numel()), s)
    else:
        rc = L.esr_tone_wgrad_recompute(_lib.ptr(Xd), _lib.ptr(dzd), _lib.ptr(keep[0][0]), _lib.ptr(keep[0][1]), _lib.ptr(keep[1][0]),
                                        0, tiles, _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gw1), _lib.ptr(gb1),
                                        _lib.ptr(scratch), C.c_int64(scratch.numel()), s)
    _lib.check(rc, "tone_wgrad")
    torch.cuda.synchronize()
    dzs = dz[:, :3].permute(0, 2, 1).reshape(tiles * 32, 3).double()
    md = mask.double()
    h = pre * md
    dH = (dzs @ W1.double()) * md
    ref = {"gw1": dzs.t() @ h, "gb1": dzs.sum(0), "gw0": dH.t() @ x.double(), "gb0": dH.sum(0)}
    got = {"gw1": gw1, "gb1": gb1, "gw0": gw0, "gb0": gb0}
    # what ONE wrong branch would cost: the largest single-pair contribution to a first-layer row, relative to the tensor's largest entry
    one = float(((dzs @ W1.double()).abs().max(0).values[:, None] * x.double().abs().max(0).values[None, :]).max() / ref["gw0"].abs().max())
    for k in ("gw0", "gb0", "gw1", "gb1"):
        e = rel_err(got[k], ref[k])
        print(f"{engine} {k}: {e:.2e} (one wrong branch on the worst pair: {one:.1e} of gw0's largest entry)")
        assert e < 2e-6, (k, e)
