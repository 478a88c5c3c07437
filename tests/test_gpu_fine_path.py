"""Fused HIP fine-stage path vs the CPU oracle (oracle/fine_path.py) and vs the
golden vectors recorded from the imported reference.  Everything goes through
the C ABI of libesr_hip.so (esr_nerf_amd.fine_engine / VoxurfF).

Tolerance: north_star's 1e-4 relative fp32, measured rel-to-max-norm
(SURVEY.md section 8(d)); discrete outputs (survivor counts) must match exactly.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
KNIFE = 1e-6      # |hidden pre-activation| below which a ReLU decision is summation-order noise (see the oracle test)


# --------------------------------------------------------------------------- helpers
def build_gpu_model(scene, seed=0, grid_seed=0, neus_alpha="interp"):
    from esr_nerf_amd.config import fine_cfg
    from esr_nerf_amd.synthetic import init_slab_model
    from esr_nerf_amd.voxurff import VoxurfF
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = VoxurfF(fine_cfg("cuda:0", neus_alpha=neus_alpha), scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min,
                scene.mask_xyz_max, scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
    init_slab_model(m, scene, seed=grid_seed)
    m.train()
    return m


def oracle_for(model, scene):
    from esr_nerf_amd.config import fine_cfg
    from oracle import fine_path as fp
    cfg = fine_cfg("cpu", neus_alpha=model.neus_alpha)
    c = fp.make_consts(cfg.app.model, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min, scene.mask_xyz_max,
                       scene.mask_alpha_init, scene.mask_density, scene.near, scene.num_voxels)
    P = fp.params_from_state_dict({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()})
    return fp, c, P


def gpu_batch(scene):
    return {k: v.cuda() for k, v in scene.batch.items()}


def run_gpu(model, scene, s_val, loss="torch", white_bg=True):
    from oracle import fine_path as fp
    b = gpu_batch(scene)
    model.zero_grad(set_to_none=True)
    res = model(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=s_val)
    out = {k: v.detach().clone() for k, v in res.items()}
    l, _ = fp.fine_loss(res, b["rgbs"], white_bg=white_bg)   # torch autograd on the device: the trainer's loss lines
    l.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return out, float(l), grads


def run_oracle(fp, c, P, scene, s_val, force=None):
    keep = {}
    res = fp.forward_training(P, c, scene.batch, s_val, keep=keep, force=force)
    l, _ = fp.fine_loss(res, scene.batch["rgbs"])
    l.backward()
    grads = {k: v.grad for k, v in P.items() if v.grad is not None}
    return {k: v.detach() for k, v in res.items()}, float(l), grads, keep


def compare(out, loss, grads, o_out, o_loss, o_grads, tol=TOL):
    for k in o_out:
        assert rel_err(out[k], o_out[k]) < tol, (k, rel_err(out[k], o_out[k]))
    assert abs(loss - o_loss) < tol * max(1.0, abs(o_loss))
    bad = {}
    for k, g in o_grads.items():
        e = rel_err(grads[k], g)
        if not e < tol:
            bad[k] = e
    assert not bad, bad
    assert set(grads) == set(o_grads)


# --------------------------------------------------------------------------- MLP engine alone
def _in_colmap(kind, row):
    if kind == 1:
        return row if row < 33 else -1
    if kind == 4:               # coarse rgbnet: colour12 | xyz PE 33 | view PE 9 | normal3 on a 72-row tile
        if row < 12: return row
        if row < 24: return -1
        if row < 27: return 54 + row - 24
        if row < 30: return 12 + row - 27
        if row < 45: return 15 + row - 30
        if row < 60: return 30 + row - 45
        if row < 69: return 45 + row - 60
        return -1
    if kind in (2, 3):          # BRDF / emission nets: colour6 | xyz PE 33 | sdf | feat24 | normal12
        if row < 6: return row
        if row == 6: return 39
        if row < 31: return 40 + row - 7
        if row < 43: return 64 + row - 31
        if row < 46: return 6 + row - 43
        if row < 61: return 9 + row - 46
        if row < 76: return 24 + row - 61
        return -1
    if row < 6: return row
    if row == 6: return 48
    if row < 31: return 49 + row - 7
    if row < 43: return 73 + row - 31
    if row < 46: return 6 + row - 43
    if row < 61: return 9 + row - 46
    if row < 76: return 24 + row - 61
    if row < 85: return 39 + row - 76
    return -1


NET = {0: dict(in_dim=85, xrows=104, nl=4, hid=192, out=3, zrows=4),
       1: dict(in_dim=33, xrows=48, nl=2, hid=192, out=3, zrows=4),
       2: dict(in_dim=76, xrows=104, nl=4, hid=128, out=5, zrows=8),
       3: dict(in_dim=76, xrows=104, nl=4, hid=128, out=3, zrows=4),
       4: dict(in_dim=57, xrows=72, nl=3, hid=128, out=3, zrows=4, cw=12)}


@pytest.mark.parametrize("kind,tiles,crow", [(0, 1, 0), (0, 37, 88), (1, 5, 0), (1, 64, 0), (2, 9, 96), (2, 300, 0),
                                             (3, 33, 88), (0, 700, 96), (4, 3, 0), (4, 130, 12)])
def test_mlp_engine_fwd_dgrad_wgrad_vs_torch(kind, tiles, crow):
    """Random tile-major inputs; reference = plain fp32 torch Linear/ReLU chain + autograd on CPU.
    Covers RadianceNet, TonemapNet, BRDFNet (5 outputs, 128 wide) and EmissionNet, and the three
    colour-row groups of the X tile."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L = eng.L
    g = torch.Generator().manual_seed(kind * 100 + tiles)
    n = NET[kind]
    in_dim, xrows, nl, hid, nout, zrows = n["in_dim"], n["xrows"], n["nl"], n["hid"], n["out"], n["zrows"]
    dims = [in_dim] + [hid] * (nl - 1) + [nout]
    Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5).requires_grad_() for i in range(nl)]
    Bs = [(torch.randn(dims[i + 1], generator=g) * 0.1).requires_grad_() for i in range(nl)]
    X = torch.randn(tiles, xrows, 32, generator=g)
    rows = [r for r in range(min(xrows, 96)) if _in_colmap(kind, r) >= 0]
    cols = [_in_colmap(kind, r) for r in rows]
    cw = n.get("cw", 6)
    src_rows = [r + crow if r < cw else r for r in rows]          # colour group actually read
    x_ref = torch.zeros(tiles * 32, in_dim)
    x_ref[:, cols] = X[:, src_rows, :].permute(0, 2, 1).reshape(tiles * 32, len(rows))
    x_ref.requires_grad_()
    h, hs = x_ref, []
    knife = torch.zeros(tiles * 32, dtype=torch.bool)
    for i in range(nl):
        h = torch.nn.functional.linear(h, Ws[i], Bs[i])
        if i + 1 < nl:
            # a pre-activation within rounding of 0 can take the other ReLU branch under the MFMA's
            # summation order; such knife-edge samples get no upstream gradient in this comparison
            knife |= (h.detach().abs() < 1e-5).any(-1)
            h = torch.relu(h)
            hs.append(h)
    dz = torch.randn(tiles * 32, nout, generator=g)
    dz[knife] = 0
    h.backward(dz)

    def tm(t, rows_):        # [tiles*32, rows] -> tile-major [tiles, rows, 32]
        return t.reshape(tiles, 32, rows_).permute(0, 2, 1).contiguous()

    packed = torch.empty(L.esr_mlp_packed_floats(kind), device="cuda")
    w = _lib.EsrMlpWeights()
    keep = [(a.detach().cuda().contiguous(), b.detach().cuda().contiguous()) for a, b in zip(Ws, Bs)]
    for i, (a, b) in enumerate(keep):
        w.w[i], w.b[i] = a.data_ptr(), b.data_ptr()
    s = _lib.stream_ptr("cuda:0")
    _lib.check(L.esr_mlp_pack(kind, C.byref(w), _lib.ptr(packed), s), "pack")
    Xd = X.cuda().contiguous()
    Hd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    Md = [torch.zeros(tiles, hid // 64, 64, dtype=torch.int32, device="cuda") for _ in range(nl - 1)]
    zout = torch.full((tiles, zrows, 32), 7.0, device="cuda")
    _lib.check(L.esr_mlp_fwd(kind, _lib.ptr(packed), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(Hd),
                             _lib.ptr_array(Md), 1, crow, _lib.ptr(zout), s), "fwd")
    assert rel_err(zout[:, :nout], tm(h.detach(), nout)) < 1e-5
    assert float(zout[:, nout:].abs().max()) == 0.0
    for a, b in zip(Hd, hs):
        assert rel_err(a, tm(b.detach(), hid)) < 1e-5
    # dgrad
    dzd = torch.zeros(tiles, zrows, 32, device="cuda")
    dzd[:, :nout] = tm(dz, nout).cuda()
    dZd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    dXd = torch.full((tiles, 64, 32), 3.0, device="cuda")
    _lib.check(L.esr_mlp_dgrad(kind, _lib.ptr(packed), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Md),
                               _lib.ptr_array(dZd), _lib.ptr(dXd), s), "dgrad")
    dx_ref = tm(x_ref.grad, in_dim)
    # dX is written for the rows that lead back to a grid, rounded up to 4 (include/esr_hip.h); the rest is untouched
    n_dx = {0: 44, 1: 36, 2: 44, 3: 44, 4: 32}[kind]
    rows_dx = [r for r in rows if r < n_dx]
    assert rel_err(dXd[:, rows_dx].cpu(), dx_ref[:, [_in_colmap(kind, r) for r in rows_dx]]) < 1e-5
    assert float((dXd[:, n_dx:] - 3.0).abs().max()) == 0.0
    # wgrad
    gw = [torch.zeros_like(w_).cuda() for w_ in Ws]
    gb = [torch.zeros_like(b).cuda() for b in Bs]
    _lib.check(L.esr_mlp_wgrad(kind, _lib.ptr(Xd), crow, _lib.ptr_array(Hd), _lib.ptr_array(dZd), _lib.ptr(dzd), 0,
                               tiles, _lib.ptr_array(gw), _lib.ptr_array(gb), _lib.ptr(eng.wgrad_scratch),
                               C.c_int64(eng.wgrad_scratch.numel()), s), "wgrad")
    for i in range(nl):
        assert rel_err(gw[i], Ws[i].grad) < 2e-5, ("gw", i, rel_err(gw[i], Ws[i].grad))
        assert rel_err(gb[i], Bs[i].grad) < 2e-5, ("gb", i)


@pytest.mark.parametrize("t_on,t_all", [(0, 5), (7, 7), (3, 11), (130, 257), (600, 1100)])
def test_merged_radiance_launches_equal_the_separate_ones(t_on, t_all):
    """esr_mlp_fwd_fine / esr_mlp_dgrad_fine (round 3: the step's three radiance forward passes as one launch, the two
    input-gradient passes as one) against the separate launches they replace -- same kernels, same per-tile arithmetic:
    every output tile, saved activation, mask, dZ and dX bit for bit."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(t_all * 7 + t_on)
    dims = [85, 192, 192, 192, 3]
    nets = {}
    for name in ("off", "emo"):
        Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5).cuda() for i in range(4)]
        Bs = [(torch.randn(dims[i + 1], generator=g) * 0.1).cuda() for i in range(4)]
        eng.pack(name, 0, Ws, Bs)
        nets[name] = (Ws, Bs)
    X = torch.randn(t_all * 104 * 32, generator=g).cuda()
    dz = torch.randn(t_all * 4 * 32, generator=g).cuda()

    def bufs():
        f = lambda rows, dt=torch.float32: torch.full((max(t_all, 1) * rows * 32,), -3, dtype=dt, device="cuda")
        return dict(H=[f(192) for _ in range(3)], M=[torch.full((max(t_all, 1) * 3 * 64,), -3, dtype=torch.int32, device="cuda") for _ in range(3)],
                    z_off=f(4), z_emo=f(4), dZ=[f(192) for _ in range(3)], dX=f(64))
    A, B = bufs(), bufs()
    pa = _lib.ptr_array
    # separate launches (round 2's sequence)
    _lib.check(L.esr_mlp_fwd_mixed(0, _lib.ptr(eng.packed["off"]), _lib.ptr(X), 0, t_on, t_all, pa(A["H"]), pa(A["M"]), 88,
                                   _lib.ptr(A["z_off"]), s), "mixed")
    _lib.check(L.esr_mlp_fwd(0, _lib.ptr(eng.packed["emo"]), _lib.ptr(X), 0, t_on, pa(A["H"]), pa(A["M"]), 1, 0,
                             _lib.ptr(A["z_emo"]), s), "emo")
    _lib.check(L.esr_mlp_dgrad(0, _lib.ptr(eng.packed["emo"]), _lib.ptr(dz), 0, t_on, pa(A["M"]), pa(A["dZ"]), _lib.ptr(A["dX"]), s), "dg emo")
    _lib.check(L.esr_mlp_dgrad(0, _lib.ptr(eng.packed["off"]), _lib.ptr(dz), t_on, t_all, pa(A["M"]), pa(A["dZ"]), _lib.ptr(A["dX"]), s), "dg off")
    # merged launches
    _lib.check(L.esr_mlp_fwd_fine(_lib.ptr(eng.packed["off"]), _lib.ptr(eng.packed["emo"]), _lib.ptr(X), t_on, t_all, pa(B["H"]),
                                  pa(B["M"]), 88, _lib.ptr(B["z_off"]), _lib.ptr(B["z_emo"]), s), "fwd_fine")
    _lib.check(L.esr_mlp_dgrad_fine(_lib.ptr(eng.packed["emo"]), _lib.ptr(eng.packed["off"]), _lib.ptr(dz), t_on, t_all, pa(B["M"]),
                                    pa(B["dZ"]), _lib.ptr(B["dX"]), s), "dgrad_fine")
    torch.cuda.synchronize()
    for k in ("z_off", "z_emo", "dX"):
        assert torch.equal(A[k], B[k]), k
    for k in ("H", "M", "dZ"):
        for l in range(3):
            assert torch.equal(A[k][l], B[k][l]), (k, l)


@pytest.mark.parametrize("t_on,t_all", [(0, 5), (7, 7), (3, 11), (130, 257), (1000, 2100)])
def test_merged_radiance_launches_bf16_equal_the_separate_ones(t_on, t_all):
    """The bf16 engine's twins (esr_mlp_fwd_fine_bf16 / esr_mlp_dgrad_fine_bf16: one workgroup = one pass's weights in
    LDS, passes share the launch's workgroups by tile count) against the separate launches: bit for bit."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0", "bf16")
    L, s = eng.L, _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(t_all * 5 + t_on)
    dims = [85, 192, 192, 192, 3]
    for name in ("off", "emo"):
        Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5).cuda() for i in range(4)]
        Bs = [(torch.randn(dims[i + 1], generator=g) * 0.1).cuda() for i in range(4)]
        eng.pack(name, 0, Ws, Bs)
    X = torch.randn(t_all * 104 * 32, generator=g).cuda()
    dz = torch.randn(t_all * 4 * 32, generator=g).cuda()
    po, pe = _lib.ptr(eng.packed["off"]), _lib.ptr(eng.packed["emo"])
    p16o, p16e = eng._p16[po.value], eng._p16[pe.value]

    def bufs():
        f = lambda rows: torch.full((max(t_all, 1) * rows * 32,), -3.0, device="cuda")
        return dict(H=[f(192) for _ in range(3)], M=[torch.full((max(t_all, 1) * 3 * 64,), -3, dtype=torch.int32, device="cuda") for _ in range(3)],
                    z_off=f(4), z_emo=f(4), dZ=[f(192) for _ in range(3)], dX=f(64))
    A, B = bufs(), bufs()
    pa = _lib.ptr_array
    _lib.check(L.esr_mlp_fwd_bf16(0, po, p16o, _lib.ptr(X), 0, t_on, pa(A["H"]), pa(A["M"]), 0, 88, _lib.ptr(A["z_off"]), s), "off det")
    _lib.check(L.esr_mlp_fwd_bf16(0, po, p16o, _lib.ptr(X), t_on, t_all, pa(A["H"]), pa(A["M"]), 1, 0, _lib.ptr(A["z_off"]), s), "off")
    _lib.check(L.esr_mlp_fwd_bf16(0, pe, p16e, _lib.ptr(X), 0, t_on, pa(A["H"]), pa(A["M"]), 1, 0, _lib.ptr(A["z_emo"]), s), "emo")
    _lib.check(L.esr_mlp_dgrad_bf16(0, p16e, _lib.ptr(dz), 0, t_on, pa(A["M"]), pa(A["dZ"]), _lib.ptr(A["dX"]), s), "dg emo")
    _lib.check(L.esr_mlp_dgrad_bf16(0, p16o, _lib.ptr(dz), t_on, t_all, pa(A["M"]), pa(A["dZ"]), _lib.ptr(A["dX"]), s), "dg off")
    _lib.check(L.esr_mlp_fwd_fine_bf16(po, p16o, pe, p16e, _lib.ptr(X), None, t_on, t_all, pa(B["H"]), pa(B["M"]), 88,
                                       _lib.ptr(B["z_off"]), _lib.ptr(B["z_emo"]), s), "fwd_fine16")
    _lib.check(L.esr_mlp_dgrad_fine_bf16(p16e, p16o, _lib.ptr(dz), t_on, t_all, pa(B["M"]), pa(B["dZ"]), _lib.ptr(B["dX"]), s), "dgrad_fine16")
    torch.cuda.synchronize()
    for k in ("z_off", "z_emo", "dX"):
        assert torch.equal(A[k], B[k]), k
    for k in ("H", "M", "dZ"):
        for l in range(3):
            assert torch.equal(A[k][l], B[k][l]), (k, l)


@pytest.mark.parametrize("tiles,t0", [(1, 0), (7, 2), (300, 0), (1500, 17)])
def test_tone_wgrad_recompute_vs_torch(tiles, t0):
    """esr_tone_wgrad_recompute (csrc/tone_wgrad.hip): the tone mapper's weight gradients from Xt and dzt alone -- the
    hidden layer recomputed in the transposed accumulator layout -- against a plain fp32 torch chain + autograd."""
    from esr_nerf_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(tiles)
    W0 = (torch.randn(192, 33, generator=g) / 33 ** 0.5).requires_grad_()
    b0 = (torch.randn(192, generator=g) * 0.1).requires_grad_()
    W1 = (torch.randn(3, 192, generator=g) / 192 ** 0.5).requires_grad_()
    b1 = (torch.randn(3, generator=g) * 0.1).requires_grad_()
    Xt = torch.randn(tiles, 48, 32, generator=g)
    Xt[:, 33:] = 7.0                                              # rows the net does not read must not matter
    x = Xt[t0:, :33].permute(0, 2, 1).reshape(-1, 33)
    pre = torch.nn.functional.linear(x, W0, b0)
    out = torch.nn.functional.linear(torch.relu(pre), W1, b1)
    dz = torch.randn(out.shape, generator=g)
    dz[(pre.detach().abs() < 1e-5).any(-1)] = 0                   # knife-edge samples: see the MLP engine test
    out.backward(dz)
    dzt = torch.zeros(tiles, 4, 32)
    dzt[t0:, :3] = dz.reshape(tiles - t0, 32, 3).permute(0, 2, 1)
    dzt[:t0] = 5.0                                                # tiles before t0 are outside the range
    dev = lambda t: t.detach().cuda().contiguous()
    gw0, gb0, gw1, gb1 = (torch.full(s_, 0.25, device="cuda") for s_ in ((192, 33), (192,), (3, 192), (3,)))
    scratch = torch.empty(L.esr_tone_wgrad_scratch_floats(), device="cuda")
    Xd, zd, W0d, b0d, W1d = dev(Xt), dev(dzt), dev(W0), dev(b0), dev(W1)        # (kept alive across the call)
    _lib.check(L.esr_tone_wgrad_recompute(_lib.ptr(Xd), _lib.ptr(zd), _lib.ptr(W0d), _lib.ptr(b0d),
                                          _lib.ptr(W1d), t0, tiles, _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gw1),
                                          _lib.ptr(gb1), _lib.ptr(scratch), C.c_int64(scratch.numel()),
                                          _lib.stream_ptr("cuda:0")), "tone_wgrad")
    for got, want in ((gw0, W0.grad), (gb0, b0.grad), (gw1, W1.grad), (gb1, b1.grad)):      # += into the outputs
        assert rel_err(got - 0.25, want) < 2e-5, rel_err(got - 0.25, want)
    # round 4: the same with the products on the 16-bit matrix cores from split fp16 planes (esr_tone_wgrad_recompute_split),
    # for output gradients of the magnitudes training sees (1e-4 .. 1e-8 per sample) as well as O(1)
    for gscale in (1.0, 1e-4):
        zs = dev(dzt * gscale * 10.0 ** (-3.0 * torch.rand(tiles, 1, 32, generator=g)) if gscale != 1.0 else dzt)
        want_s = None
        if gscale != 1.0:
            for p_ in (W0, b0, W1, b1):
                p_.grad = None
            pre2 = torch.nn.functional.linear(x, W0, b0)
            out2 = torch.nn.functional.linear(torch.relu(pre2), W1, b1)
            dz2 = zs[t0:, :3].cpu().permute(0, 2, 1).reshape(-1, 3)
            out2.backward(dz2)
        want_s = (W0.grad, b0.grad, W1.grad, b1.grad)
        amax = torch.zeros(1, device="cuda")
        _lib.check(L.esr_absmax(_lib.ptr(zs[t0:]), C.c_int64((tiles - t0) * 128), _lib.ptr(amax), _lib.stream_ptr("cuda:0")), "absmax")
        hw0, hb0, hw1, hb1 = (torch.zeros(s_, device="cuda") for s_ in ((192, 33), (192,), (3, 192), (3,)))
        _lib.check(L.esr_tone_wgrad_recompute_split(_lib.ptr(Xd), _lib.ptr(zs), _lib.ptr(W0d), _lib.ptr(b0d), _lib.ptr(W1d),
                                                    _lib.ptr(amax), t0, tiles, _lib.ptr(hw0), _lib.ptr(hb0), _lib.ptr(hw1),
                                                    _lib.ptr(hb1), _lib.ptr(scratch), C.c_int64(scratch.numel()),
                                                    _lib.stream_ptr("cuda:0")), "tone_wgrad_split")
        for got, want in zip((hw0, hb0, hw1, hb1), want_s):
            assert rel_err(got, want) < 2e-5, (gscale, rel_err(got, want))


@pytest.mark.parametrize("tiles,t0", [(1, 0), (7, 2), (300, 0), (1500, 17)])
def test_tone_wgrad_recompute_bf16_vs_emulation(tiles, t0):
    """esr_tone_wgrad_recompute_bf16: the same kernel scheme with bf16 matrix operands, against a torch emulation of
    exactly that arithmetic -- Xt, W0, dzt, W1 and the recomputed dZt rounded to bf16 where they become a matrix operand,
    everything the kernel sums on the vector lanes (dW1, db0, the 33rd input column) unrounded, sums in fp32.  2e-3 of the max-norm, as for the bf16 MLP
    kernels (a value on a bf16 rounding boundary may round the other way under a different summation order)."""
    from esr_nerf_amd import _lib
    L = _lib.lib()
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    g = torch.Generator().manual_seed(tiles + 99)
    W0 = torch.randn(192, 33, generator=g) / 33 ** 0.5
    b0 = torch.randn(192, generator=g) * 0.1
    W1 = torch.randn(3, 192, generator=g) / 192 ** 0.5
    Xt = torch.randn(tiles, 48, 32, generator=g)
    Xt[:, 33:] = 0.0                                              # (tone_in_fwd writes zeros there)
    x = Xt[t0:, :33].permute(0, 2, 1).reshape(-1, 33)
    pre = bf(x) @ bf(W0).t() + b0
    ht = torch.relu(pre)
    dz = torch.randn(x.shape[0], 3, generator=g)
    dz[(pre.abs() < 1e-3).any(-1)] = 0                            # knife-edge samples (bf16 operands: a wider edge)
    dZt = (bf(dz) @ bf(W1)) * (ht > 0)                            # fp32 accumulators of the kernel's dHt MFMA, masked
    gw0 = torch.cat([bf(dZt).t() @ bf(x[:, :32]),                 # columns 0..31: bf16 MFMA operands
                     (dZt.t() @ x[:, 32:33])], 1)                 # column 32, db0, dW1: fp32 vector sums of unrounded values
    want = (gw0, dZt.sum(0), dz.t() @ ht, dz.sum(0))
    dzt = torch.zeros(tiles, 4, 32)
    dzt[t0:, :3] = dz.reshape(tiles - t0, 32, 3).permute(0, 2, 1)
    dzt[:t0] = 5.0
    dev = lambda t: t.detach().cuda().contiguous()
    gw0, gb0, gw1, gb1 = (torch.full(s_, 0.25, device="cuda") for s_ in ((192, 33), (192,), (3, 192), (3,)))
    scratch = torch.empty(L.esr_tone_wgrad_scratch_floats(), device="cuda")
    Xd, zd, W0d, b0d, W1d = dev(Xt), dev(dzt), dev(W0), dev(b0), dev(W1)
    _lib.check(L.esr_tone_wgrad_recompute_bf16(_lib.ptr(Xd), _lib.ptr(zd), _lib.ptr(W0d), _lib.ptr(b0d),
                                               _lib.ptr(W1d), t0, tiles, _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gw1),
                                               _lib.ptr(gb1), _lib.ptr(scratch), C.c_int64(scratch.numel()),
                                               _lib.stream_ptr("cuda:0")), "tone_wgrad16")
    for name, got, w in zip(("gw0", "gb0", "gw1", "gb1"), (gw0, gb0, gw1, gb1), want):
        assert rel_err(got - 0.25, w) < 2e-3, (name, rel_err(got - 0.25, w))


def test_loss_kernel_matches_trainer_loss():
    from esr_nerf_amd.fine_engine import FineEngine
    from oracle import fine_path as fp
    eng = FineEngine("cuda:0")
    g = torch.Generator().manual_seed(3)
    n = 777
    res = {"etc/alphainv_cum": torch.rand(n, generator=g).requires_grad_(),
           "srgb/rgb": (torch.rand(n, 3, generator=g) * 1.3 - 0.1).requires_grad_(),
           "lin/rgb": (torch.rand(n, 3, generator=g) * 1.6 - 0.2).requires_grad_()}
    res["etc/white_bg"] = res["etc/alphainv_cum"][..., None]
    rgbs = torch.rand(n, 3, generator=g)
    rgbs[::7] = 1.0
    l, _ = fp.fine_loss(res, rgbs)
    l.backward()
    loss, g_last, g_srgb, g_lin = eng.loss_fwd_bwd(res["etc/alphainv_cum"].detach().cuda(),
                                                   res["srgb/rgb"].detach().cuda().contiguous(),
                                                   res["lin/rgb"].detach().cuda().contiguous(), rgbs.cuda())
    assert abs(float(loss) - float(l)) < 1e-6
    assert rel_err(g_srgb, res["srgb/rgb"].grad) < 1e-5
    assert rel_err(g_lin, res["lin/rgb"].grad) < 1e-5
    assert rel_err(g_last, res["etc/alphainv_cum"].grad) < 1e-5


# --------------------------------------------------------------------------- end to end
def test_golden_reference_vectors(golden_case, golden_params):
    """Fixtures generated by the IMPORTED reference: outputs, loss and all 23 gradients."""
    from conftest import golden_alpha_mode, golden_scene
    name, z = golden_case
    sd, _ = golden_params
    sc = golden_scene(name, z)
    for k in ("rays_o", "rays_d", "viewdirs", "em_modes", "rgbs"):
        assert torch.equal(sc.batch[k], z["in/" + k]), k            # the generator is deterministic
    m = build_gpu_model(sc, neus_alpha=golden_alpha_mode(name))
    m.load_state_dict({k: v.cuda() for k, v in sd.items()})
    assert m.off_color.grid.is_contiguous(memory_format=torch.channels_last_3d)
    out, loss, grads = run_gpu(m, sc, float(z["in/s_val"]), white_bg=bool(z["in/white_bg"]))
    for k in out:
        assert rel_err(out[k], z["out/" + k]) < TOL, (k, rel_err(out[k], z["out/" + k]))
    assert abs(loss - float(z["loss"])) < 1e-5
    bad = {k[5:]: rel_err(grads[k[5:]], v) for k, v in z.items()
           if k.startswith("grad/") and not rel_err(grads[k[5:]], v) < TOL}
    assert not bad, bad
    # discrete: survivors after the in-box test and before compositing
    assert m.last_counts["m0"] == int((~z["native/sample/mask_outbbox"]).sum())
    assert m.last_counts["m1"] == int(z["native/mask_keep"].sum())             # MaskCache.forward of the reference
    assert m.last_counts["m2"] == z["native/a2w/alpha"].numel()
    if "_prune" in name:
        lc = m.last_counts
        assert lc["m0"] > lc["m1"] >= lc["m2"] >= lc["m3"] > 0 and lc["m1"] < 0.7 * lc["m0"]
    # FineStep (fused loss kernel, what bench.py times) on the same fixture, incl. the white_bg = False variant
    from esr_nerf_amd.trainer import FineStep
    loss2, grads2 = FineStep(m, white_bg=bool(z["in/white_bg"])).forward_loss_backward(gpu_batch(sc), float(z["in/s_val"]))
    assert abs(float(loss2) - float(z["loss"])) < 1e-5
    bad = {k[5:]: rel_err(grads2[k[5:]], v) for k, v in z.items()
           if k.startswith("grad/") and not rel_err(grads2[k[5:]], v) < TOL}
    assert not bad, bad


@pytest.mark.parametrize("name,oblique,s_val,n_rays,mask,alpha", [
    ("tiny", False, 20.0, None, "full", "interp"), ("tiny", True, 90.0, 200, "full", "interp"),
    ("small", False, 220.0, None, "full", "interp"), ("small", True, 45.0, 300, "full", "interp"),
    ("tiny", True, 400.0, 1, "full", "interp"),
    # pruning mask cache: exact M0 > M1 > M2 > M3 against the oracle (module.py:78-114, voxurff.py:189-191)
    ("tiny", False, 20.0, None, "prune", "interp"), ("tiny", True, 90.0, 200, "prune", "interp"),
    ("small", False, 220.0, None, "prune", "interp"), ("small", True, 45.0, 300, "prune", "interp"),
    # cfg neus_alpha: "grad" (functions.py:45-69): 7 taps per march sample, backward through all of them;
    # oblique rays clip the box faces (clamped taps, shortened index distance)
    ("tiny", True, 90.0, 200, "prune", "grad"), ("small", True, 45.0, 300, "full", "grad"),
    ("small", False, 220.0, None, "prune", "grad"),
])
def test_fused_path_vs_oracle(name, oblique, s_val, n_rays, mask, alpha):
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene(name, s_val=s_val, oblique=oblique, n_rays=n_rays, seed=3, mask=mask)
    m = build_gpu_model(sc, seed=1, grid_seed=2, neus_alpha=alpha)
    fp, c, P = oracle_for(m, sc)
    out, loss, grads = run_gpu(m, sc, s_val)
    # A hidden unit whose pre-activation lies within fp32 summation noise of 0 (~1e-7 here) may take the other ReLU
    # branch on the GPU: both are correct roundings, but the sample's gradient jumps (one such unit in ~10 k samples
    # moved emo_color.grid's gradient by 3e-3 when feat_fwd's taps changed by nothing but an fma).  Round 4 dropped the
    # rays holding such samples and ran both sides again; now NOTHING is dropped: the oracle takes over the HIP step's
    # discrete decisions (tests/decisions.py: survivor set, ReLU branches), and every decision it would have taken
    # differently is arbitrated in float64 -- it must sit on its boundary (assert_legitimate).
    from decisions import assert_legitimate, hip_decisions
    fp.FLIP_LOG = []
    try:
        o_out, o_loss, o_grads, keep = run_oracle(fp, c, P, sc, s_val, force=hip_decisions(m))
        assert_legitimate(keep, fp.FLIP_LOG, what=f"{name}/{mask}/{alpha}")
    finally:
        fp.FLIP_LOG = None
    n0, n1, n2, n3 = keep["counts"]
    lc = m.last_counts
    assert (lc["m0"], lc["m1"], lc["m2"], lc["m3"]) == (n0, n1, n2, n3)
    if mask == "prune":
        assert n0 > n1 >= n2 >= n3 > 0 and n1 < 0.7 * n0
        if s_val > 20.0:
            assert n1 > n2 > n3
        # survivors of a ray are non-contiguous steps: the NeuS neighbour rule pairs samples across the gaps
        rid, sid = keep["ray_id"], keep["step_id"]
        assert int(((rid[1:] == rid[:-1]) & (sid[1:] - sid[:-1] > 1)).sum()) > 5
    compare(out, loss, grads, o_out, o_loss, o_grads)


def test_all_rays_miss_or_empty():
    """Degenerate batches: rays that never enter the box -> zero colour, alphainv_last = 1, zero grads."""
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("tiny", n_rays=16)
    sc.batch["rays_o"] = sc.batch["rays_o"] + torch.tensor([10.0, 0.0, 0.0])
    m = build_gpu_model(sc)
    out, loss, grads = run_gpu(m, sc, 20.0)
    assert torch.equal(out["etc/alphainv_cum"].cpu(), torch.ones(16))
    assert float(out["srgb/rgb"].abs().max()) == 0.0 and m.last_counts["m3"] == 0
    assert float(grads["sdf.grid"].abs().max()) == 0.0


def test_fused_loss_path_equals_autograd_path():
    """bench.py drives engine.loss_fwd_bwd + engine.backward directly; same numbers as autograd."""
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=256)
    m = build_gpu_model(sc)
    out, loss, grads = run_gpu(m, sc, 60.0)
    step = FineStep(m)
    b = gpu_batch(sc)
    loss2, grads2 = step.forward_loss_backward(b, 60.0)
    assert abs(float(loss2) - loss) < 1e-6
    for k, g in grads.items():
        assert rel_err(grads2[k], g) < 1e-5, k


def test_c2_full_size_properties():
    """BASELINE config C2 (4096 rays x 128 samples): size-independent properties.
    (a) the slab gives exactly 128 samples per ray and, at s_val=20, all survive;
    (b) rays are independent: a 48-ray subset evaluated ALONE by the CPU oracle matches the
        same rays inside the full GPU batch;
    (c) two runs agree (float atomics only reorder last bits)."""
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("C2", s_val=20.0)
    m = build_gpu_model(sc)
    out, loss, grads = run_gpu(m, sc, 20.0)
    lc = m.last_counts
    assert lc["m0"] == lc["m1"] == lc["m2"] == lc["m3"] == 4096 * 128
    out2, loss2, grads2 = run_gpu(m, sc, 20.0)
    assert rel_err(out2["srgb/rgb"], out["srgb/rgb"]) < 1e-6 and abs(loss - loss2) < 1e-6
    assert rel_err(grads2["sdf.grid"], grads["sdf.grid"]) < 1e-5
    idx = torch.arange(0, 4096, 4096 // 48)[:48]
    sub = slab_scene("C2", s_val=20.0)
    sub.batch = {k: v[idx].contiguous() for k, v in sc.batch.items()}
    fp, c, P = oracle_for(m, sub)
    with torch.no_grad():
        res = fp.forward_training(P, c, sub.batch, 20.0)
    for k in ("etc/alphainv_cum", "srgb/rgb", "lin/rgb"):
        assert rel_err(out[k][idx.cuda()], res[k]) < TOL, k


@pytest.mark.parametrize("mask", ["full", "prune"])
def test_forward_evaluate_golden_and_psnr(mask):
    """Image rendering (VoxurfF.forward_evaluate) against the reference-generated fixture, all 12 result keys
    for both emissive modes; PSNR between the two renderings of the slab 'image' far beyond the 0.1 dB bar."""
    from conftest import load_npz
    from esr_nerf_amd.synthetic import slab_scene
    z = {k: torch.from_numpy(np.asarray(v)) for k, v in load_npz("fine_g16_eval.npz" if mask == "full" else "fine_g16_eval_prune.npz").items()}
    sd = {k: torch.from_numpy(v) for k, v in load_npz("fine_g16_params.npz").items() if not k.startswith("__")}
    sc = slab_scene("g16", s_val=60.0, oblique=True, mask=mask)
    m = build_gpu_model(sc)
    m.load_state_dict({k: v.cuda() for k, v in sd.items()})
    m.s_val = 60.0
    m.eval()
    for em in (0, 1):
        res = m(rays_o=z["in/rays_o"].cuda(), rays_d=z["in/rays_d"].cuda(), viewdirs=z["in/viewdirs"].cuda(),
                em_modes=em, pos_rt=z["in/pos_rt"].cuda())
        keys = [k[5:] for k in z if k.startswith(f"out{em}/")]
        assert set(keys) == set(res) and len(keys) == 12
        for k in keys:
            assert res[k].shape == z[f"out{em}/{k}"].shape, k
            assert rel_err(res[k], z[f"out{em}/{k}"]) < TOL, (em, k, rel_err(res[k], z[f"out{em}/{k}"]))
        ref = (z[f"out{em}/srgb/rgb"] + z[f"out{em}/etc/white_bg"]).clamp(0, 1)
        got = (res["srgb/rgb"].cpu() + res["etc/white_bg"].cpu()).clamp(0, 1)
        mse = float(((ref - got) ** 2).mean())
        assert mse < 1e-10                                           # PSNR > 100 dB between the two renderings
    m.train()
    assert m.forward == m.forward_training


def test_forward_evaluate_all_miss():
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("tiny", n_rays=16)
    m = build_gpu_model(sc)
    m.eval()
    o = (sc.batch["rays_o"] + torch.tensor([10.0, 0.0, 0.0])).cuda()
    res = m(rays_o=o, rays_d=sc.batch["rays_d"].cuda(), viewdirs=sc.batch["viewdirs"].cuda(), em_modes=1,
            pos_rt=torch.eye(3).cuda())
    assert float(res["srgb/rgb"].abs().max()) == 0.0 and torch.equal(res["etc/white_bg"].cpu(), torch.ones(16, 1))
    assert rel_err(res["etc/disp"], torch.full((16,), 1.0 / sc.far)) < 1e-6


@pytest.mark.parametrize("kind,tiles,crow", [(0, 5, 0), (0, 300, 88), (1, 70, 0), (2, 33, 96), (3, 9, 88), (4, 40, 12)])
def test_mlp_engine_bf16_vs_emulation(kind, tiles, crow):
    """bf16-operand / fp32-accumulate MLP kernels (fwd, dgrad, wgrad) against a torch emulation of exactly that
    arithmetic: weights rounded to bf16 once, activations / gradients rounded where they become an MFMA operand,
    sums in fp32.  Tolerance 2e-3 of the max-norm (a value that sits on a bf16 rounding boundary may round the
    other way under a different fp32 summation order)."""
    from esr_nerf_amd import _lib
    from esr_nerf_amd.fine_engine import FineEngine
    eng = FineEngine("cuda:0")
    L = eng.L
    g = torch.Generator().manual_seed(kind * 77 + tiles)
    n = NET[kind]
    in_dim, xrows, nl, hid, nout, zrows = n["in_dim"], n["xrows"], n["nl"], n["hid"], n["out"], n["zrows"]
    cw = n.get("cw", 6)
    dims = [in_dim] + [hid] * (nl - 1) + [nout]
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(nl)]
    Bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(nl)]
    X = torch.randn(tiles, xrows, 32, generator=g)
    rows = [r for r in range(min(xrows, 96)) if _in_colmap(kind, r) >= 0]
    cols = [_in_colmap(kind, r) for r in rows]
    src_rows = [r + crow if r < cw else r for r in rows]
    x = torch.zeros(tiles * 32, in_dim)
    x[:, cols] = X[:, src_rows, :].permute(0, 2, 1).reshape(tiles * 32, len(rows))
    bf = lambda t: t.bfloat16().float()
    Wb = [bf(w) for w in Ws]
    hs, pre, a = [], [], x
    for i in range(nl):
        zz = bf(a) @ Wb[i].T + Bs[i]
        if i + 1 < nl:
            pre.append(zz)
            a = torch.relu(zz)
            hs.append(a)
    z_ref = zz
    def tm(t, rows_):
        return t.reshape(tiles, 32, rows_).permute(0, 2, 1).contiguous()

    s = _lib.stream_ptr("cuda:0")
    packed = torch.empty(L.esr_mlp_packed_floats(kind), device="cuda")
    packed16 = torch.empty(L.esr_mlp_packed_bf16_elems(kind), dtype=torch.bfloat16, device="cuda")
    w = _lib.EsrMlpWeights()
    keep = [(a_.cuda().contiguous(), b_.cuda().contiguous()) for a_, b_ in zip(Ws, Bs)]
    for i, (a_, b_) in enumerate(keep):
        w.w[i], w.b[i] = a_.data_ptr(), b_.data_ptr()
    _lib.check(L.esr_mlp_pack(kind, C.byref(w), _lib.ptr(packed), s), "pack")
    _lib.check(L.esr_mlp_pack_bf16(kind, C.byref(w), _lib.ptr(packed16), s), "pack16")
    Xd = X.cuda().contiguous()
    Hd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    Md = [torch.zeros(tiles, hid // 64, 64, dtype=torch.int32, device="cuda") for _ in range(nl - 1)]
    zout = torch.full((tiles, zrows, 32), 7.0, device="cuda")
    _lib.check(L.esr_mlp_fwd_bf16(kind, _lib.ptr(packed), _lib.ptr(packed16), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(Hd),
                                  _lib.ptr_array(Md), 1, crow, _lib.ptr(zout), s), "fwd16")
    T16 = 2e-3
    # saved tiles are bf16 in the row-quad layout [row / 4][32 sample slots][4 rows] (mlp_common.h: store_tiles_bf16)
    # with the 32 sample slots of a quad ordered slot(s) = 8 ((s >> 1) & 3) + 2 (s >> 3) + (s & 1)
    slot = torch.tensor([8 * ((s_ >> 1) & 3) + 2 * (s_ >> 3) + (s_ & 1) for s_ in range(32)], device="cuda")
    as_bf16 = lambda t: (t.view(-1).view(torch.bfloat16)[: tiles * hid * 32].view(tiles, hid // 4, 32, 4)[:, :, slot, :]
                         .permute(0, 1, 3, 2).reshape(tiles, hid, 32).float())
    assert rel_err(zout[:, :nout], tm(z_ref, nout)) < T16
    for a_, b_ in zip(Hd, hs):                 # one bf16 ulp (2^-8) where the two fp32 values straddle a rounding boundary
        assert rel_err(as_bf16(a_), tm(bf(b_), hid)) < 5e-3
    dz = torch.randn(tiles * 32, nout, generator=g)
    knife = torch.zeros(tiles * 32, dtype=torch.bool)
    for p_, h_ in zip(pre, Hd):
        knife |= (p_.abs() < 1e-4).any(-1)
        # ... and samples where the kernel's ReLU decision differs from the emulation's: one bf16 ulp in an upstream
        # activation (a rounding boundary crossed by the fp32 summation order) moves a downstream pre-activation by ~1e-3
        hk = as_bf16(h_).cpu().permute(0, 2, 1).reshape(tiles * 32, hid)
        knife |= ((hk > 0) != (p_ > 0)).any(-1)
    assert float(knife.float().mean()) < 0.2
    dz[knife] = 0
    # backward emulation
    gW, gB, dA = [None] * nl, [None] * nl, dz
    dZs = [None] * (nl - 1)
    for i in range(nl - 1, -1, -1):
        inp = x if i == 0 else hs[i - 1]
        gW[i] = bf(dA).T @ bf(inp)
        gB[i] = (dA if i == nl - 1 else bf(dA)).sum(0)          # hidden-layer dZ is stored (and summed) as bf16
        d_in = bf(dA) @ Wb[i]
        if i > 0:
            dA = d_in * (pre[i - 1] > 0)
            dZs[i - 1] = dA
        else:
            dx_ref = d_in

    dzd = torch.zeros(tiles, zrows, 32, device="cuda")
    dzd[:, :nout] = tm(dz, nout).cuda()
    dZd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    dXd = torch.full((tiles, 64, 32), 3.0, device="cuda")
    _lib.check(L.esr_mlp_dgrad_bf16(kind, _lib.ptr(packed16), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Md),
                                    _lib.ptr_array(dZd), _lib.ptr(dXd), s), "dgrad16")
    for a_, b_ in zip(dZd, dZs):
        assert rel_err(as_bf16(a_), tm(bf(b_), hid)) < 5e-3
    rows64 = [r for r in rows if r < 64]
    assert rel_err(dXd[:, rows64].cpu(), tm(dx_ref, in_dim)[:, [_in_colmap(kind, r) for r in rows64]]) < T16
    gw = [torch.zeros_like(w_).cuda() for w_ in Ws]
    gb = [torch.zeros_like(b_).cuda() for b_ in Bs]
    _lib.check(L.esr_mlp_wgrad_bf16(kind, _lib.ptr(Xd), crow, _lib.ptr_array(Hd), _lib.ptr_array(dZd), _lib.ptr(dzd), 0,
                                    tiles, _lib.ptr_array(gw), _lib.ptr_array(gb), _lib.ptr(eng.wgrad_scratch),
                                    C.c_int64(eng.wgrad_scratch.numel()), s), "wgrad16")
    for i in range(nl):
        assert rel_err(gw[i], gW[i]) < T16, ("gw", i, rel_err(gw[i], gW[i]))
        assert rel_err(gb[i], gB[i]) < T16, ("gb", i, rel_err(gb[i], gB[i]))
    if kind == 0:
        # a NULL dZ[l] in the input-gradient pass: that layer's tile is computed but not stored, everything else is unchanged
        dZn = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
        dXn = torch.full((tiles, 64, 32), 3.0, device="cuda")
        _lib.check(L.esr_mlp_dgrad_bf16(kind, _lib.ptr(packed16), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Md),
                                        _lib.ptr_array([dZn[0], dZn[1], None]), _lib.ptr(dXn), s), "dgrad16 without dZ[2]")
        assert torch.equal(dXn, dXd) and torch.equal(dZn[0], dZd[0]) and torch.equal(dZn[1], dZd[1])


def test_bf16_mode_end_to_end_close_to_fp32_and_psnr():
    """mlp_dtype="bf16" (bf16 MFMA operands in the MLPs, everything else fp32): training outputs / loss /
    gradients stay close to the fp32 path, and the rendered image agrees with the fp32 rendering far inside the
    0.1 dB PSNR bar BASELINE.json sets for the bf16 configurations."""
    import math
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=512, seed=9)
    m32 = build_gpu_model(sc, seed=1, grid_seed=2)
    m16 = build_gpu_model(sc, seed=1, grid_seed=2)
    m16.mlp_dtype = "bf16"
    out32, loss32, g32 = run_gpu(m32, sc, 60.0)
    out16, loss16, g16 = run_gpu(m16, sc, 60.0)
    assert m16.engine.bf16 and not m32.engine.bf16
    assert m16.last_counts == m32.last_counts                      # the march is fp32 in both
    for k in out32:
        assert rel_err(out16[k], out32[k]) < 1e-2, (k, rel_err(out16[k], out32[k]))
    assert abs(loss16 - loss32) < 2e-3 * max(1.0, abs(loss32))
    for k in g32:        # bf16 operands flip ReLU units whose pre-activation is within ~1e-3 of zero: per-element
        a_, b_ = g16[k].flatten().double(), g32[k].flatten().double()        # noise of a few %, same direction
        cos = float((a_ * b_).sum() / (a_.norm() * b_.norm()).clamp_min(1e-30))
        assert cos > 0.995 and rel_err(g16[k], g32[k]) < 0.2, (k, cos, rel_err(g16[k], g32[k]))
    b = gpu_batch(sc)
    for m in (m32, m16):
        m.s_val = 60.0
        m.eval()
    kw = dict(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=1, pos_rt=torch.eye(3).cuda())
    r32, r16 = m32(**kw), m16(**kw)
    img = lambda r: (r["srgb/rgb"] + r["etc/white_bg"]).clamp(0, 1)
    gt = b["rgbs"]
    psnr = lambda x: -10.0 * math.log10(float(((x - gt) ** 2).mean()))
    assert abs(psnr(img(r32)) - psnr(img(r16))) < 0.1
    mse = float(((img(r32) - img(r16)) ** 2).mean())
    assert -10.0 * math.log10(max(mse, 1e-12)) > 45.0               # bf16 vs fp32 rendering: > 45 dB


def test_pack_batch_equals_single_packs():
    """esr_mlp_pack_batch (every net of a step in one launch, fp32 buffers + their bf16 twins) against esr_mlp_pack /
    esr_mlp_pack_bf16 net by net: bit-identical packed buffers for all five net kinds."""
    import ctypes as C
    from esr_nerf_amd import _lib
    L, s = _lib.lib(), _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(12)
    dims = {0: (85, 192, 3, 3), 1: (33, 192, 1, 3), 2: (76, 128, 3, 5), 3: (76, 128, 3, 3), 4: (57, 128, 2, 3)}
    kinds = [0, 0, 1, 2, 3, 4]
    ws, keep, p32a, p32b, p16a, p16b = [], [], [], [], [], []
    for kind in kinds:
        i, h, nh, o = dims[kind]
        sizes = [(h, i)] + [(h, h)] * (nh - 1) + [(o, h)]
        w = _lib.EsrMlpWeights()
        for l, (r, c) in enumerate(sizes):
            wt, bt = torch.randn(r, c, generator=g).cuda(), torch.randn(r, generator=g).cuda()
            keep += [wt, bt]
            w.w[l], w.b[l] = wt.data_ptr(), bt.data_ptr()
        ws.append(w)
        n32, n16 = L.esr_mlp_packed_floats(kind), L.esr_mlp_packed_bf16_elems(kind)
        p32a.append(torch.full((n32,), 7.0, device="cuda")); p32b.append(torch.full((n32,), 9.0, device="cuda"))
        p16a.append(torch.full((n16,), 7.0, dtype=torch.bfloat16, device="cuda"))
        p16b.append(torch.full((n16,), 9.0, dtype=torch.bfloat16, device="cuda"))
        _lib.check(L.esr_mlp_pack(kind, C.byref(w), _lib.ptr(p32a[-1]), s), "pack")
        _lib.check(L.esr_mlp_pack_bf16(kind, C.byref(w), _lib.ptr(p16a[-1]), s), "pack16")
    n = len(kinds)
    ka = (C.c_int32 * n)(*kinds)
    wa = (C.c_void_p * n)(*[C.addressof(w) for w in ws])
    a32 = (C.c_void_p * n)(*[t.data_ptr() for t in p32b])
    a16 = (C.c_void_p * n)(*[t.data_ptr() for t in p16b])
    _lib.check(L.esr_mlp_pack_batch(n, ka, wa, a32, a16, None, s), "pack_batch")
    for k in range(n):
        assert torch.equal(p32a[k], p32b[k]), kinds[k]
        assert torch.equal(p16a[k].view(torch.int16), p16b[k].view(torch.int16)), kinds[k]
    # fp32 only (packed16 NULL) leaves the bf16 buffers alone
    for t in p32b:
        t.fill_(3.0)
    _lib.check(L.esr_mlp_pack_batch(n, ka, wa, a32, None, None, s), "pack_batch")
    assert all(torch.equal(x, y) for x, y in zip(p32a, p32b))
    assert L.esr_mlp_pack_batch(9, ka, wa, a32, a16, None, s) != 0
