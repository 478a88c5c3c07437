"""C oracle (oracle/esr_oracle.c) against an independent numpy restatement and
against the native-op traffic recorded from the imported reference run."""
import numpy as np
import pytest
import torch

from oracle import native


def np_sample(o, d, bmin, bmax, near, far, stepdist):
    """Independent float32 numpy statement of the sampler
    (render_utils_kernel.cu:12-79,144-242)."""
    f = np.float32
    o, d = o.astype(f), d.astype(f)
    v = np.where(d == 0, f(1e-6), d).astype(f)
    a = ((bmax - o) / v).astype(f)
    b = ((bmin - o) / v).astype(f)
    tmin = np.minimum(a, b).max(-1)
    tmax = np.maximum(a, b).min(-1)
    tmin = np.maximum(np.minimum(tmin, f(far)), f(near)).astype(f)
    tmax = np.maximum(np.minimum(tmax, f(far)), f(near)).astype(f)
    sq = (d[:, 0] * d[:, 0]).astype(f)
    sq = (sq + (d[:, 1] * d[:, 1]).astype(f)).astype(f)
    sq = (sq + (d[:, 2] * d[:, 2]).astype(f)).astype(f)
    nrm = np.sqrt(sq).astype(f)
    ln = (((tmax - tmin).astype(f) * nrm).astype(f) / f(stepdist)).astype(f)
    n = np.maximum(np.ceil(ln).astype(np.float64), 1.0).astype(np.int64)
    start = (o + (d * tmin[:, None]).astype(f)).astype(f)
    dr = (d / nrm[:, None]).astype(f)
    ray_id = np.repeat(np.arange(len(n)), n)
    step = np.concatenate([np.arange(k) for k in n]) if len(n) else np.zeros(0, np.int64)
    dist = (f(stepdist) * step.astype(f)).astype(f)
    pts = (start[ray_id] + (dr[ray_id] * dist[:, None]).astype(f)).astype(f)
    out = ((bmin > pts) | (bmax < pts)).any(-1)
    return pts, out, ray_id, step, n, tmin, tmax


def rand_rays(n, seed, zero_axis=True):
    g = np.random.default_rng(seed)
    o = g.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    d = g.normal(size=(n, 3)).astype(np.float32) * g.uniform(0.2, 3, (n, 1)).astype(np.float32)
    if zero_axis and n > 4:
        d[::5, g.integers(0, 3)] = 0.0
    return o, d


@pytest.mark.parametrize("n,seed", [(1, 0), (7, 1), (300, 2), (1000, 3)])
def test_sampler_matches_numpy(n, seed):
    o, d = rand_rays(n, seed)
    bmin = np.array([-1, -0.8, -0.5], np.float32)
    bmax = np.array([1, 0.9, 0.25], np.float32)
    near, far, sd = 0.05, 1e9, 0.0123
    ref = np_sample(o, d, bmin, bmax, near, far, sd)
    got = native.sample_pts_on_rays(torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(bmin),
                                    torch.from_numpy(bmax), near, far, sd)
    names = ["pts", "mask", "ray_id", "step_id", "n_steps", "t_min", "t_max"]
    for nm, r, g in zip(names, ref, got):
        assert np.array_equal(np.asarray(r), g.numpy()), nm      # bit-exact, floats included


def test_sampler_empty_and_missing_rays():
    e = torch.zeros(0, 3)
    out = native.sample_pts_on_rays(e, e, torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), 0.1, 1e9, 0.01)
    assert out[0].shape == (0, 3) and out[4].numel() == 0
    # a ray that misses the box still gets >= 1 (out-of-box) sample
    o = torch.tensor([[5.0, 5.0, 5.0]])
    d = torch.tensor([[0.0, 0.0, 1.0]])
    out = native.sample_pts_on_rays(o, d, torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3), 0.1, 1e9, 0.01)
    assert out[4].item() == 1 and bool(out[1].all())


def np_a2w(alpha, ray_id, n_rays):
    w = np.zeros_like(alpha); T = np.ones_like(alpha)
    last = np.ones(n_rays, np.float32)
    i_s = np.zeros(n_rays, np.int64); i_e = np.zeros(n_rays, np.int64)
    for r in np.unique(ray_id):
        idx = np.nonzero(ray_id == r)[0]
        i_s[r] = idx[0]
        tc = np.float32(1.0)
        stop = idx[-1] + 1
        for i in idx:
            T[i] = tc
            w[i] = np.float32(tc * alpha[i])
            tc = np.float32(np.float64(tc) * (1.0 - np.float64(alpha[i])))
            if tc < 1e-3:
                stop = i + 1
                break
        i_e[r] = stop
        last[r] = tc
    return w, T, last, i_s, i_e


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_alpha2weight_fwd_bwd(seed):
    g = np.random.default_rng(seed)
    n_rays = 40
    counts = g.integers(0, 30, n_rays)
    counts[3] = 0; counts[-1] = 0
    ray_id = np.repeat(np.arange(n_rays), counts)
    alpha = g.uniform(0, 1, len(ray_id)).astype(np.float32) ** 3
    alpha[g.uniform(size=len(alpha)) < 0.1] = 0.9999       # force early stops
    ref = np_a2w(alpha, ray_id, n_rays)
    got = native.alpha2weight(torch.from_numpy(alpha), torch.from_numpy(ray_id), n_rays)
    for r, t in zip(ref, got):
        assert np.array_equal(r, t.numpy())
    # backward against the analytic reverse scan in float64 (tolerance) ...
    gw = g.normal(size=len(alpha)).astype(np.float32)
    gl = g.normal(size=n_rays).astype(np.float32)
    grad = native.alpha2weight_backward(*[torch.from_numpy(np.asarray(x)) for x in (alpha, *ref)], n_rays,
                                        torch.from_numpy(gw), torch.from_numpy(gl)).numpy()
    w, T, last, i_s, i_e = ref
    exp = np.zeros_like(alpha, dtype=np.float64)
    for r in range(n_rays):
        back = np.float64(gl[r]) * last[r]
        for i in range(i_e[r] - 1, i_s[r] - 1, -1):
            exp[i] = gw[i] * np.float64(T[i]) - back / (1 - np.float64(alpha[i]) + 1e-10)
            back += np.float64(gw[i]) * w[i]
    assert np.allclose(grad, exp, rtol=2e-4, atol=1e-5)
    # ... samples after the early stop get exactly zero
    for r in range(n_rays):
        seg = np.nonzero(ray_id == r)[0]
        assert np.all(grad[seg[seg >= i_e[r]]] == 0)


def test_alpha2weight_empty():
    out = native.alpha2weight(torch.zeros(0), torch.zeros(0, dtype=torch.int64), 5)
    assert out[0].numel() == 0 and torch.equal(out[2], torch.ones(5))


def np_tv(param, grad, wy, wz, dense):
    p = param[0, 0].astype(np.float32)
    add = np.zeros_like(p)
    def cl(x):
        return np.clip(x, -1, 1).astype(np.float32)
    wy6, wz6 = np.float32(np.float32(wy) / 6), np.float32(np.float32(wz) / 6)
    # accumulate in the same order: -k, +k, -j, +j, -i, +i
    add[:, :, 1:] += wz6 * cl(p[:, :, 1:] - p[:, :, :-1])
    add[:, :, :-1] += wz6 * cl(p[:, :, :-1] - p[:, :, 1:])
    add[:, 1:, :] += wy6 * cl(p[:, 1:] - p[:, :-1])
    add[:, :-1, :] += wy6 * cl(p[:, :-1] - p[:, 1:])
    add[1:] += wz6 * cl(p[1:] - p[:-1])
    add[:-1] += wz6 * cl(p[:-1] - p[1:])
    out = grad.copy()
    m = np.ones_like(p, bool) if dense else (grad[0, 0] != 0)
    out[0, 0][m] += add[m]
    return out


@pytest.mark.parametrize("dense", [True, False])
def test_tv_add_grad(dense):
    g = np.random.default_rng(5)
    param = (g.normal(size=(1, 1, 7, 6, 5)) * 1.5).astype(np.float32)
    grad = g.normal(size=(1, 1, 7, 6, 5)).astype(np.float32)
    grad[g.uniform(size=grad.shape) < 0.4] = 0
    exp = np_tv(param, grad, 0.3, 0.7, dense)      # wx is unused by the reference kernel
    got = torch.from_numpy(grad.copy())
    native.total_variation_add_grad(torch.from_numpy(param), got, 123.0, 0.3, 0.7, dense)
    assert np.allclose(got.numpy(), exp, rtol=1e-6, atol=1e-6)
    if not dense:
        assert np.array_equal(got.numpy()[grad == 0], grad[grad == 0])


def test_segment_sum():
    g = np.random.default_rng(0)
    idx = np.sort(g.integers(0, 9, 50))
    src = g.normal(size=(50, 3)).astype(np.float32)
    out = torch.zeros(10, 3)
    native.segment_sum(torch.from_numpy(src), torch.from_numpy(idx), out)
    exp = np.zeros((10, 3), np.float32)
    np.add.at(exp, idx, src)
    assert np.allclose(out.numpy(), exp, atol=1e-5)


def test_native_ops_reproduce_recorded_reference_traffic(golden_case):
    """The fixture holds what the imported reference model passed to / got from the
    native ops; the C oracle must reproduce it bit for bit on this machine too."""
    name, z = golden_case
    from esr_nerf_amd.synthetic import slab_scene
    sc = slab_scene("g16")
    got = native.sample_pts_on_rays(z["in/rays_o"], z["in/rays_d"], sc.xyz_min, sc.xyz_max,
                                    float(z["native/sample/near"]), 1e9, float(z["native/sample/stepdist"]))
    for nm, t in zip(["ray_pts", "mask_outbbox", "ray_id", "step_id", "N_steps", "t_min", "t_max"], got):
        assert torch.equal(t, z["native/sample/" + nm]), nm
    out = native.alpha2weight(z["native/a2w/alpha"], z["native/a2w/ray_id"], z["in/rays_o"].shape[0])
    for nm, t in zip(["weight", "T", "alphainv_last", "i_start", "i_end"], out):
        assert torch.equal(t, z["native/a2w/" + nm]), nm
    g = native.alpha2weight_backward(z["native/a2w/alpha"], *out, z["in/rays_o"].shape[0],
                                     z["native/a2wb/grad_weights"], z["native/a2wb/grad_last"])
    assert torch.equal(g, z["native/a2wb/grad"])


# ---- the ops the reference's modules export but its Python never calls (no reference traffic, no golden vectors exist for
# them): the C restatements against independent numpy / torch statements of the kernels' formulas
def test_dead_ray_helpers_match_the_sampler_statement():
    """infer_t_minmax / infer_n_samples / infer_ray_start_dir (render_utils_kernel.cu:12-79) are the three steps
    sample_pts_on_rays runs internally: same numbers as np_sample's, bit for bit."""
    o, d = rand_rays(300, 4)
    bmin, bmax = np.array([-1, -1, -0.25], np.float32), np.array([1, 1, 0.25], np.float32)
    _, _, _, _, n_ref, tmin_ref, tmax_ref = np_sample(o, d, bmin, bmax, 0.2, 6.0, 0.01)
    to = torch.from_numpy
    t_min, t_max = native.infer_t_minmax(to(o), to(d), to(bmin), to(bmax), 0.2, 6.0)
    assert np.array_equal(t_min.numpy(), tmin_ref) and np.array_equal(t_max.numpy(), tmax_ref)
    n = native.infer_n_samples(to(d), t_min, t_max, 0.01)
    assert n.dtype == torch.int64 and np.array_equal(n.numpy(), n_ref)
    start, dirs = native.infer_ray_start_dir(to(o), to(d), t_min)
    f = np.float32
    sq = ((d[:, 0] * d[:, 0]).astype(f) + (d[:, 1] * d[:, 1]).astype(f)).astype(f)
    nrm = np.sqrt((sq + (d[:, 2] * d[:, 2]).astype(f)).astype(f)).astype(f)
    assert np.array_equal(start.numpy(), (o + (d * tmin_ref[:, None]).astype(f)).astype(f))
    assert np.array_equal(dirs.numpy(), (d / nrm[:, None]).astype(f))


def test_dead_ndc_and_background_samplers():
    """sample_ndc_pts_on_rays (render_utils_kernel.cu:243-269): o + d step / (N - 1), exact in numpy float32.
    sample_bg_pts_on_rays (:301-340) against the torch lines its source quotes as the original implementation."""
    g = np.random.default_rng(3)
    o = g.uniform(-0.3, 0.3, (40, 3)).astype(np.float32)
    d = g.normal(size=(40, 3)).astype(np.float32)
    bmin, bmax = np.array([-1, -1, -1], np.float32), np.array([1, 1, 0.5], np.float32)
    to = torch.from_numpy
    S = 17
    pts, mask = native.sample_ndc_pts_on_rays(to(o), to(d), to(bmin), to(bmax), S)
    dist = (np.arange(S, dtype=np.float32) / np.float32(S - 1)).astype(np.float32)
    ref = (o[:, None, :] + (d[:, None, :] * dist[None, :, None]).astype(np.float32)).astype(np.float32)
    assert pts.shape == (40, S, 3) and np.array_equal(pts.numpy(), ref)
    assert mask.dtype == torch.bool and np.array_equal(mask.numpy(), ((bmin > ref) | (bmax < ref)).any(-1))
    assert mask.any() and not mask.all()
    # background: the quoted torch implementation, in float64 (the kernel mixes float and double; it agrees to fp32 rounding)
    t_max = torch.from_numpy(g.uniform(1.0, 2.0, 40).astype(np.float32))
    N, bg = 12, 0.3
    got = native.sample_bg_pts_on_rays(to(o), to(d), t_max, bg, N)
    ro, rd, tm = to(o).double(), to(d).double(), t_max.double()
    ori_t_outer = tm[:, None] - 1 + 1 / torch.linspace(1, 0, N + 1, dtype=torch.float64)[:-1]
    ori = (ro[:, None, :] + rd[:, None, :] * ori_t_outer[:, :, None]).reshape(-1, 3)
    t_outer = ori.norm(dim=-1)
    R_outer = t_outer / ori.abs().amax(1)
    o2i = R_outer.pow(2) / t_outer.pow(2) * (1 - bg) + R_outer / t_outer * bg
    ref = (ori * o2i[:, None]).reshape(40, N, 3)
    assert got.shape == (40, N, 3)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_dead_maskcache_lookup():
    """render_utils_kernel.cu:366-392: nearest voxel (round half away from zero) of a bool volume; outside reads False."""
    g = torch.Generator().manual_seed(1)
    world = torch.rand(5, 6, 7, generator=g) < 0.5
    xyz = torch.rand(500, 3, generator=g) * 3.0 - 1.0
    xyz[0] = torch.tensor([0.5 / 2.0, 0.0, 0.0])           # a tie: 0.5 rounds to 1, not to 0
    scale, shift = torch.tensor([2.0, 2.5, 3.0]), torch.tensor([0.0, 0.5, 1.0])
    got = native.maskcache_lookup(world, xyz, scale, shift)
    ijk = (xyz * scale + shift)
    r = torch.where(ijk >= 0, torch.floor(ijk + 0.5), torch.ceil(ijk - 0.5)).long()
    inside = ((r >= 0) & (r < torch.tensor([5, 6, 7]))).all(-1)
    ref = torch.zeros(500, dtype=torch.bool)
    ref[inside] = world[r[inside, 0], r[inside, 1], r[inside, 2]]
    assert got.dtype == torch.bool and torch.equal(got, ref)
    assert bool(inside.any()) and not bool(inside.all()) and bool(got.any())


@pytest.mark.parametrize("nonuni", [False, True])
def test_dead_raw2alpha_and_its_backward(nonuni):
    """render_utils_kernel.cu:431-460,504-530: alpha = 1 - (1 + exp(d + shift))^(-interval) and the gradient the kernel
    states, min(e, 1e10) (1 + e)^(-interval - 1) interval g -- which is autograd's gradient of that alpha where e is finite."""
    g = torch.Generator().manual_seed(2)
    n = 4000
    dens = torch.randn(n, generator=g) * 6.0
    dens[:3] = torch.tensor([95.0, -95.0, 0.0])            # e = inf (fp32 exp overflows at 88.7), e = 0+, e = e^shift
    shift = -1.5
    iv_t = torch.rand(n, generator=g) * 0.9 + 0.1
    iv = iv_t if nonuni else 0.37
    fwd = native.raw2alpha_nonuni if nonuni else native.raw2alpha
    bwd = native.raw2alpha_nonuni_backward if nonuni else native.raw2alpha_backward
    e, a = fwd(dens, shift, iv)
    d64 = (dens + shift).double().requires_grad_(True)        # the kernel rounds density + shift to fp32 before the exponential
    e64 = torch.exp(d64)
    iv64 = iv_t.double() if nonuni else iv
    a64 = 1 - (1 + e64) ** (-iv64)
    fin = torch.isfinite(e)
    assert bool((~fin).any()) and float(a[~fin].min()) == 1.0                    # (1 + inf)^(-interval) = 0
    assert float((e[fin].double() - e64.detach()[fin]).abs().div(e64.detach()[fin].clamp_min(1e-30)).max()) < 5e-7    # (e^-96.5 is a denormal)
    assert float((a.double() - a64.detach()).abs().max()) < 3e-7
    gb = torch.randn(n, generator=g)
    got = bwd(e, gb, iv)
    a64.backward(gb.double())
    ok = fin & (e < 1e10)
    assert float((got[ok].double() - d64.grad[ok]).abs().max() / d64.grad[ok].abs().max()) < 1e-6
    assert bool(torch.isfinite(got[fin]).all())
    # e = inf: min(e, 1e10) * (1 + inf)^(..) = 1e10 * 0 = 0 -- the kernel's clamp is what keeps inf * 0 = NaN out
    assert float(got[~fin].abs().max()) == 0.0


@pytest.mark.parametrize("dense", [True, False])
def test_dead_masked_tv_add_grad(dense):
    """total_variation_kernel.cu:38-66: the live kernel's stencil with every term times mask[cell] mask[neighbour], wx on the
    fastest axis (the live one uses wz there) -- against a torch statement with shifted slices."""
    g = torch.Generator().manual_seed(7)
    p = torch.randn(1, 2, 5, 6, 7, generator=g) * 1.5
    m = (torch.rand(1, 2, 5, 6, 7, generator=g) < 0.7).float()
    grad = torch.randn(1, 2, 5, 6, 7, generator=g)
    if not dense:
        grad[torch.rand(grad.shape, generator=g) < 0.5] = 0.0
    g0 = grad.clone()
    wx, wy, wz = 0.6, 1.2, 2.4
    native.total_variation_add_grad_new(p, grad, m, wx, wy, wz, dense)
    add = torch.zeros_like(p, dtype=torch.float64)
    for axis, w in ((4, wx), (3, wy), (2, wz)):
        for sgn in (-1, 1):
            nb, mb = torch.roll(p, sgn, axis).double(), torch.roll(m, sgn, axis).double()
            term = (w / 6) * (p.double() - nb).clamp(-1, 1) * m.double() * mb
            idx = torch.arange(p.shape[axis])
            edge = (idx == 0) if sgn == 1 else (idx == p.shape[axis] - 1)       # roll(+1) brings index-1: no left neighbour at 0
            shape = [1] * 5
            shape[axis] = -1
            add += torch.where(edge.view(shape), torch.zeros_like(term), term)
    ref = g0.double() + (add if dense else add * (g0 != 0))
    assert float((grad.double() - ref).abs().max()) < 1e-6
    assert bool(((grad - g0) != 0).any())


# ---- the reference's DOUBLE instantiation of the three live ops: float locals inside double kernels
def np_sample_f64(o, d, bmin, bmax, near, far, stepdist):
    """numpy statement of render_utils_kernel.cu:12-79,167-194 with scalar_t = double: every `float` local of the kernels is an
    explicit astype(float32), everything else float64."""
    f, D = np.float32, np.float64
    o, d, bmin, bmax = o.astype(D), d.astype(D), bmin.astype(D), bmax.astype(D)
    v = np.where(d == 0, D(1e-6), d).astype(f)
    a = ((bmax - o) / v.astype(D)).astype(f)
    b = ((bmin - o) / v.astype(D)).astype(f)
    tmin = np.maximum(np.minimum(np.minimum(a, b).max(-1), f(far)), f(near)).astype(f)
    tmax = np.maximum(np.minimum(np.maximum(a, b).min(-1), f(far)), f(near)).astype(f)
    nrm = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(f)
    ln = (tmax.astype(D) - tmin.astype(D)) * nrm.astype(D) / D(f(stepdist))
    n = np.maximum(np.ceil(ln), 1.0).astype(np.int64)
    start = o + d * tmin.astype(D)[:, None]
    dr = d / nrm.astype(D)[:, None]
    ray_id = np.repeat(np.arange(len(n)), n)
    step = np.concatenate([np.arange(k) for k in n]) if len(n) else np.zeros(0, np.int64)
    dist = (f(stepdist) * step.astype(f)).astype(f)
    pts = (start[ray_id] + dr[ray_id] * dist.astype(D)[:, None]).astype(f)
    out = ((bmin > pts.astype(D)) | (bmax < pts.astype(D))).any(-1)
    return pts.astype(D), out, ray_id, step, n, tmin.astype(D), tmax.astype(D)


@pytest.mark.parametrize("n,seed", [(1, 0), (300, 2), (3000, 5)])
def test_sampler_double_instantiation(n, seed):
    o, d = rand_rays(n, seed)
    g = np.random.default_rng(seed + 100)
    o = o.astype(np.float64) + g.normal(size=o.shape) * 1e-9          # values that are NOT floats: the roundings matter
    d = d.astype(np.float64) * (1 + g.normal(size=d.shape) * 1e-9)
    bmin, bmax = np.array([-1, -0.8, -0.5]), np.array([1, 0.9, 0.25])
    ref = np_sample_f64(o, d, bmin, bmax, 0.05, 1e9, 0.0123)
    to = torch.from_numpy
    got = native.sample_pts_on_rays(to(o), to(d), to(bmin), to(bmax), 0.05, 1e9, 0.0123)
    assert got[0].dtype == torch.float64 and got[5].dtype == torch.float64 and got[4].dtype == torch.int64
    for nm, r, t in zip(["pts", "mask", "ray_id", "step_id", "n_steps", "t_min", "t_max"], ref, got):
        assert np.array_equal(np.asarray(r), t.numpy()), nm
    # float-valued although stored as doubles (the kernels' float locals)
    assert np.array_equal(got[0].numpy(), got[0].numpy().astype(np.float32).astype(np.float64))
    # and close to, but not the same as, the fp32 instantiation on the rounded inputs
    f32 = native.sample_pts_on_rays(to(o).float(), to(d).float(), to(bmin).float(), to(bmax).float(), 0.05, 1e9, 0.0123)
    assert abs(int(f32[4].sum()) - int(got[4].sum())) <= max(2, n // 100)


@pytest.mark.parametrize("seed", [0, 1])
def test_alpha2weight_double_instantiation(seed):
    g = np.random.default_rng(seed)
    n_rays = 60
    counts = g.integers(0, 40, n_rays)
    counts[3] = 0
    ray_id = np.repeat(np.arange(n_rays), counts)
    alpha = g.uniform(0, 1, len(ray_id)) ** 3
    alpha[g.uniform(size=len(alpha)) < 0.1] = 0.9999
    f, D = np.float32, np.float64
    w, T = np.zeros(len(alpha)), np.ones(len(alpha))
    last, i_s, i_e = np.ones(n_rays), np.zeros(n_rays, np.int64), np.zeros(n_rays, np.int64)
    for r in range(n_rays):
        seg = np.nonzero(ray_id == r)[0]
        if len(seg):
            i_s[r], i_e[r] = seg[0], seg[-1] + 1
    # (rays without samples keep 0 / 0; a ray FOLLOWING an empty one still starts where its samples do)
    for r in range(n_rays):
        tc, i = f(1.0), i_s[r]
        while i < i_e[r]:
            T[i] = D(tc)
            w[i] = D(tc) * alpha[i]
            tc = f(D(tc) * (1.0 - alpha[i]))                     # `float T_cum` in a double kernel
            i += 1
            if D(tc) < 1e-3:
                break
        if len(np.nonzero(ray_id == r)[0]):
            i_e[r] = i
        last[r] = D(tc)
    to = torch.from_numpy
    got = native.alpha2weight(to(alpha), to(ray_id), n_rays)
    assert got[0].dtype == torch.float64
    assert np.array_equal(got[0].numpy(), w) and np.array_equal(got[1].numpy(), T) and np.array_equal(got[2].numpy(), last)
    assert np.array_equal(got[4].numpy(), i_e)
    gw, gl = g.normal(size=len(alpha)), g.normal(size=n_rays)
    grad = native.alpha2weight_backward(to(alpha), *got, n_rays, to(gw), to(gl)).numpy()
    exp = np.zeros(len(alpha))
    for r in range(n_rays):
        back = f(gl[r] * last[r])                                 # `float back_cum`
        for i in range(i_e[r] - 1, i_s[r] - 1, -1):
            exp[i] = gw[i] * T[i] - D(back) / ((1.0 - alpha[i]) + 1e-10)
            back = f(D(back) + gw[i] * w[i])
    assert grad.dtype == np.float64 and np.array_equal(grad, exp)


def test_what_if_the_reference_binary_contracts_multiply_adds():
    """DESIGN.md section 3, stated limit (ii): the reference's kernels are built with nvcc's default contraction, the oracle (and
    sampler.hip) round every operation separately, and the real binary cannot be run here.  The oracle's what-if variant fuses
    EVERY multiply-add of the sampler's statements; on the ray sets of the BASELINE configurations this is how far such a build
    can be: NO ray changes its step count, a sample point moves by at most an ulp of its box coordinate, and the only discrete
    output that changes is the out-of-box flag of a ray's ENTRY sample, which lies ON the box face (distance exactly 0) -- a
    sample in empty space in front of the mask cache, so nothing behind the sampler sees it."""
    from esr_nerf_amd.config import fine_cfg
    from esr_nerf_amd.synthetic import slab_scene
    from oracle import fine_path as fp
    seen_flip = False
    for name, kw in (("C2", dict(s_val=20.0)), ("C2", dict(s_val=20.0, oblique=True)), ("C4", dict(s_val=220.0))):
        sc = slab_scene(name, **kw)
        c = fp.make_consts(fine_cfg("cpu").app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                           sc.mask_alpha_init, sc.mask_density, sc.near, sc.num_voxels)
        sd = float(c.stepsize * c.voxel_size)
        b = sc.batch
        a = native.sample_pts_on_rays(b["rays_o"], b["rays_d"], sc.xyz_min, sc.xyz_max, sc.near, 1e9, sd)
        f = native.sample_pts_on_rays_fma(b["rays_o"], b["rays_d"], sc.xyz_min, sc.xyz_max, sc.near, 1e9, sd)
        assert torch.equal(a[4], f[2]), name                                  # step counts: identical
        assert float((a[0] - f[0]).abs().max()) <= 2.4e-7                     # <= one ulp of a coordinate of size <= 2
        flip = (a[1] != f[1]).nonzero()[:, 0]
        if flip.numel():
            seen_flip = True
            assert bool((a[3][flip] == 0).all())                              # entry samples only
            p = a[0][flip]
            on_face = torch.minimum((p - sc.xyz_min).abs(), (p - sc.xyz_max).abs()).min(-1).values
            assert float(on_face.max()) == 0.0
            assert flip.numel() < 0.003 * a[1].numel()
    assert seen_flip                                                          # (the oblique set: ~1000 of 406 k samples)
