"""Coarse stage (VoxurfC) on the HIP path: dense operators, the fused renderer
against the reference-generated golden vectors and against oracle/coarse_path.py.  Tolerance 1e-4."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_npz, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("dims", [(9, 7, 6), (32, 32, 8), (3, 2, 5), (40, 33, 17)])
def test_dense_operators_and_adjoints(dims):
    """Gaussian smoothing (replicate padding) and the dense central-difference gradient vs torch, and their
    gather-form adjoints vs autograd."""
    from esr_nerf_amd import _lib
    from oracle import coarse_path as cp
    L = _lib.lib()
    s = _lib.stream_ptr("cuda:0")
    g = torch.Generator().manual_seed(sum(dims))
    grid = torch.randn(1, 1, *dims, generator=g).requires_grad_(True)
    ker = cp.gaussian_kernel(5, 0.8)
    kw = (C.c_float * 125)(*ker.flatten().tolist())
    sm = cp.smooth_grid(grid, ker)
    gd = grid.detach()[0, 0].contiguous().cuda()
    out = torch.empty_like(gd)
    _lib.check(L.esr_gauss3d_fwd(_lib.ptr(gd), kw, 5, *dims, _lib.ptr(out), s), "gauss fwd")
    assert rel_err(out, sm[0, 0]) < 1e-6
    go = torch.randn(*dims, generator=g)
    sm.backward(go[None, None])
    gin = torch.full(dims, 0.5, device="cuda")                       # the adjoint ADDS into its output
    god = go.cuda().contiguous()
    _lib.check(L.esr_gauss3d_bwd(_lib.ptr(god), kw, 5, *dims, _lib.ptr(gin), s), "gauss bwd")
    assert rel_err(gin - 0.5, grid.grad[0, 0]) < 2e-6
    grid.grad = None
    dg = cp.dense_gradient(grid, 0.0625)
    cg = torch.empty(*dims, 3, device="cuda")
    _lib.check(L.esr_central_grad_fwd(_lib.ptr(gd), *dims, C.c_float(0.0625), _lib.ptr(cg), s), "cgrad fwd")
    assert torch.equal(cg.cpu(), dg[0].permute(1, 2, 3, 0).detach())            # bit-exact
    gg = torch.randn(*dims, 3, generator=g)
    dg.backward(gg.permute(3, 0, 1, 2)[None])
    gs = torch.full(dims, -1.0, device="cuda")
    ggd = gg.cuda().contiguous()
    _lib.check(L.esr_central_grad_bwd(_lib.ptr(ggd), *dims, C.c_float(0.0625), _lib.ptr(gs), s), "cgrad bwd")
    assert rel_err(gs + 1.0, grid.grad[0, 0]) < 2e-6
