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
