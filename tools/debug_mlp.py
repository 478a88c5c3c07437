import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fine_path import NET, _in_colmap
from esr_nerf_amd import _lib
from esr_nerf_amd.fine_engine import FineEngine
def run(kind, tiles, crow):
    eng = FineEngine("cuda:0"); L = eng.L
    g = torch.Generator().manual_seed(kind * 100 + tiles)
    n = NET[kind]
    in_dim, xrows, nl, hid, nout, zrows = n["in_dim"], n["xrows"], n["nl"], n["hid"], n["out"], n["zrows"]
    dims = [in_dim] + [hid] * (nl - 1) + [nout]
    Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5).requires_grad_() for i in range(nl)]
    Bs = [(torch.randn(dims[i + 1], generator=g) * 0.1).requires_grad_() for i in range(nl)]
    X = torch.randn(tiles, xrows, 32, generator=g)
    rows = [r for r in range(min(xrows, 96)) if _in_colmap(kind, r) >= 0]
    cols = [_in_colmap(kind, r) for r in rows]
    src_rows = [r + crow if r < 6 else r for r in rows]
    x_ref = torch.zeros(tiles * 32, in_dim)
    x_ref[:, cols] = X[:, src_rows, :].permute(0, 2, 1).reshape(tiles * 32, len(rows))
    x_ref.requires_grad_()
    h, hs = x_ref, []
    for i in range(nl):
        h = torch.nn.functional.linear(h, Ws[i], Bs[i])
        if i + 1 < nl: h = torch.relu(h); hs.append(h)
    dz = torch.randn(tiles * 32, nout, generator=g)
    h.backward(dz)
    tm = lambda t, r: t.reshape(tiles, 32, r).permute(0, 2, 1).contiguous()
    packed = torch.empty(L.esr_mlp_packed_floats(kind), device="cuda")
    w = _lib.EsrMlpWeights()
    keep = [(a.detach().cuda().contiguous(), b.detach().cuda().contiguous()) for a, b in zip(Ws, Bs)]
    for i, (a, b) in enumerate(keep): w.w[i], w.b[i] = a.data_ptr(), b.data_ptr()
    s = _lib.stream_ptr("cuda:0")
    _lib.check(L.esr_mlp_pack(kind, C.byref(w), _lib.ptr(packed), s), "pack")
    Xd = X.cuda().contiguous()
    Hd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    Md = [torch.zeros(tiles, hid // 64, 64, dtype=torch.int32, device="cuda") for _ in range(nl - 1)]
    zout = torch.full((tiles, zrows, 32), 7.0, device="cuda")
    _lib.check(L.esr_mlp_fwd(kind, _lib.ptr(packed), _lib.ptr(Xd), 0, tiles, _lib.ptr_array(Hd), _lib.ptr_array(Md), 1, crow, _lib.ptr(zout), s), "fwd")
    dzd = torch.zeros(tiles, zrows, 32, device="cuda"); dzd[:, :nout] = tm(dz, nout).cuda()
    dZd = [torch.zeros(tiles, hid, 32, device="cuda") for _ in range(nl - 1)]
    dXd = torch.zeros(tiles, 64, 32, device="cuda")
    _lib.check(L.esr_mlp_dgrad(kind, _lib.ptr(packed), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(Md), _lib.ptr_array(dZd), _lib.ptr(dXd), s), "dgrad")
    dx_ref = tm(x_ref.grad, in_dim)
    rows64 = [r for r in rows if r < 64]
    err = (dXd[:, rows64].cpu() - dx_ref[:, [_in_colmap(kind, r) for r in rows64]]).abs()
    print("kind", kind, tiles, crow, "dX err max", float(err.max()), "ref max", float(dx_ref.abs().max()))
    print("  per-row max err (rows64 idx):", [(rows64[i], round(float(err[:, i].max()), 4)) for i in range(len(rows64)) if err[:, i].max() > 1e-4][:20])
    bad_tiles = (err.amax(dim=(1, 2)) > 1e-4).nonzero().flatten().tolist()
    print("  bad tiles:", bad_tiles[:20], len(bad_tiles))
    # dZ check
    gw = [torch.zeros_like(w_).cuda() for w_ in Ws]; gb = [torch.zeros_like(b).cuda() for b in Bs]
    _lib.check(L.esr_mlp_wgrad(kind, _lib.ptr(Xd), crow, _lib.ptr_array(Hd), _lib.ptr_array(dZd), _lib.ptr(dzd), 0, tiles, _lib.ptr_array(gw), _lib.ptr_array(gb), _lib.ptr(eng.wgrad_scratch), C.c_int64(eng.wgrad_scratch.numel()), s), "wgrad")
    for i in range(nl):
        e = (gw[i].cpu() - Ws[i].grad).abs()
        print("  gw", i, "max err", float(e.max()), "ref max", float(Ws[i].grad.abs().max()))
        if i == 0:
            colerr = e.amax(0)
            print("    bad cols:", [(c, round(float(colerr[c]), 3)) for c in range(in_dim) if colerr[c] > 1e-3][:30])
for a in [(2, 300, 0), (0, 37, 88), (2, 9, 96)]:
    run(*a)
