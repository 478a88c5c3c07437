"""ctypes front-end of the C oracle (oracle/esr_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (esr_nerf_amd) never imports it.

Every function mirrors one op of the reference's pybind modules
(app/utils/base/cuda/render_utils.cpp:170-184, total_variation.cpp:29-32) on
CPU torch tensors, so it can also be plugged into the imported reference
(oracle/gen_golden.py) in place of the CUDA extension.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libesr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "esr_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _sample_pts_on_rays_f64(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """The reference's double instantiation (float locals inside: esr_oracle.c)."""
    c = lambda t: t.detach().to(torch.float64).contiguous()
    rays_o, rays_d, xyz_min, xyz_max = c(rays_o), c(rays_d), c(xyz_min), c(xyz_max)
    n = rays_o.shape[0]
    t_min, t_max = torch.empty(n, dtype=torch.float64), torch.empty(n, dtype=torch.float64)
    n_steps = torch.empty(n, dtype=torch.int64)
    L = lib()
    L.esr_oracle_sample_count_f64(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), ctypes.c_float(float(near)),
                                  ctypes.c_float(float(far)), ctypes.c_float(float(stepdist)), ctypes.c_int64(n),
                                  _p(t_min), _p(t_max), _p(n_steps))
    total = int(n_steps.sum().item())
    ray_pts = torch.empty(total, 3, dtype=torch.float64)
    mask = torch.empty(total, dtype=torch.uint8)
    ray_id, step_id = torch.empty(total, dtype=torch.int64), torch.empty(total, dtype=torch.int64)
    L.esr_oracle_sample_fill_f64(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _p(t_min), _p(n_steps),
                                 ctypes.c_float(float(stepdist)), ctypes.c_int64(n), _p(ray_pts), _p(mask), _p(ray_id), _p(step_id))
    return [ray_pts, mask.bool(), ray_id, step_id, n_steps, t_min, t_max]


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> [ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max]
    (render_utils_kernel.cu:196-242)."""
    if rays_o.dtype == torch.float64:
        return _sample_pts_on_rays_f64(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist)
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    xyz_min, xyz_max = _f32(xyz_min), _f32(xyz_max)
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=torch.float32)
    t_max = torch.empty(n, dtype=torch.float32)
    n_steps = torch.empty(n, dtype=torch.int64)
    L = lib()
    L.esr_oracle_sample_count(
        _p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
        ctypes.c_float(float(near)), ctypes.c_float(float(far)),
        ctypes.c_float(float(stepdist)), ctypes.c_int64(n),
        _p(t_min), _p(t_max), _p(n_steps),
    )
    total = int(n_steps.sum().item())
    ray_pts = torch.empty(total, 3, dtype=torch.float32)
    mask = torch.empty(total, dtype=torch.uint8)
    ray_id = torch.empty(total, dtype=torch.int64)
    step_id = torch.empty(total, dtype=torch.int64)
    L.esr_oracle_sample_fill(
        _p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _p(t_min), _p(n_steps),
        ctypes.c_float(float(stepdist)), ctypes.c_int64(n),
        _p(ray_pts), _p(mask), _p(ray_id), _p(step_id),
    )
    return [ray_pts, mask.bool(), ray_id, step_id, n_steps, t_min, t_max]


def alpha2weight(alpha, ray_id, n_rays):
    """-> [weight, T, alphainv_last, i_start, i_end] (render_utils_kernel.cu:619-651)."""
    if alpha.dtype == torch.float64:
        alpha = alpha.detach().contiguous()
        ray_id = ray_id.to(torch.int64).contiguous()
        m = alpha.shape[0]
        weight, T = torch.empty(m, dtype=torch.float64), torch.empty(m, dtype=torch.float64)
        last = torch.empty(n_rays, dtype=torch.float64)
        i_s, i_e = torch.empty(n_rays, dtype=torch.int64), torch.empty(n_rays, dtype=torch.int64)
        lib().esr_oracle_alpha2weight_f64(_p(alpha), _p(ray_id), ctypes.c_int64(m), ctypes.c_int64(int(n_rays)),
                                          _p(weight), _p(T), _p(last), _p(i_s), _p(i_e))
        return [weight, T, last, i_s, i_e]
    alpha = _f32(alpha)
    ray_id = ray_id.to(torch.int64).contiguous()
    m = alpha.shape[0]
    weight = torch.empty(m, dtype=torch.float32)
    T = torch.empty(m, dtype=torch.float32)
    last = torch.empty(n_rays, dtype=torch.float32)
    i_s = torch.empty(n_rays, dtype=torch.int64)
    i_e = torch.empty(n_rays, dtype=torch.int64)
    lib().esr_oracle_alpha2weight(
        _p(alpha), _p(ray_id), ctypes.c_int64(m), ctypes.c_int64(int(n_rays)),
        _p(weight), _p(T), _p(last), _p(i_s), _p(i_e),
    )
    return [weight, T, last, i_s, i_e]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays,
                          grad_weights, grad_last):
    """-> grad wrt alpha (render_utils_kernel.cu:679-707)."""
    if alpha.dtype == torch.float64:
        c = lambda t: t.detach().to(torch.float64).contiguous()
        alpha, weight, T, alphainv_last, gw, gl = (c(t) for t in (alpha, weight, T, alphainv_last, grad_weights, grad_last))
        grad = torch.empty(alpha.shape[0], dtype=torch.float64)
        lib().esr_oracle_alpha2weight_backward_f64(_p(alpha), _p(weight), _p(T), _p(alphainv_last), _p(i_start.contiguous()),
                                                   _p(i_end.contiguous()), ctypes.c_int64(alpha.shape[0]), ctypes.c_int64(int(n_rays)),
                                                   _p(gw), _p(gl), _p(grad))
        return grad
    alpha, weight, T = _f32(alpha), _f32(weight), _f32(T)
    alphainv_last = _f32(alphainv_last)
    gw, gl = _f32(grad_weights), _f32(grad_last)
    m = alpha.shape[0]
    grad = torch.empty(m, dtype=torch.float32)
    lib().esr_oracle_alpha2weight_backward(
        _p(alpha), _p(weight), _p(T), _p(alphainv_last),
        _p(i_start.contiguous()), _p(i_end.contiguous()),
        ctypes.c_int64(m), ctypes.c_int64(int(n_rays)), _p(gw), _p(gl), _p(grad),
    )
    return grad


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """In-place on `grad` (total_variation_kernel.cu:68-98)."""
    assert param.is_contiguous() and grad.is_contiguous()
    assert param.dtype == torch.float32 and grad.dtype == torch.float32
    lib().esr_oracle_tv_add_grad(
        _p(param.detach()), _p(grad),
        ctypes.c_float(float(wx)), ctypes.c_float(float(wy)), ctypes.c_float(float(wz)),
        ctypes.c_int64(param.shape[2]), ctypes.c_int64(param.shape[3]),
        ctypes.c_int64(param.shape[4]), ctypes.c_int64(param.numel()),
        ctypes.c_int(1 if dense_mode else 0),
    )


def segment_sum(src, index, out):
    """torch_scatter.segment_coo(src, index, out=out, reduce='sum') for sorted index."""
    src = _f32(src)
    index = index.to(torch.int64).contiguous()
    assert out.is_contiguous() and out.dtype == torch.float32
    c = 1 if src.dim() == 1 else src.shape[1]
    lib().esr_oracle_segment_sum(
        _p(src), _p(index), ctypes.c_int64(src.shape[0]), ctypes.c_int64(c), _p(out)
    )
    return out


# ---- the exported-but-never-called ops of the two modules (render_utils.cpp:171-173,175-181, total_variation.cpp:31) ----
def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    """-> [t_min, t_max] (render_utils_kernel.cu:12-35,82-103)."""
    rays_o, rays_d, xyz_min, xyz_max = _f32(rays_o), _f32(rays_d), _f32(xyz_min), _f32(xyz_max)
    n = rays_o.shape[0]
    t_min, t_max = torch.empty(n, dtype=torch.float32), torch.empty(n, dtype=torch.float32)
    unused = torch.empty(n, dtype=torch.int64)             # (the step counts of that statement, at stepdist 1: not returned)
    lib().esr_oracle_sample_count(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), ctypes.c_float(float(near)),
                                  ctypes.c_float(float(far)), ctypes.c_float(1.0), ctypes.c_int64(n), _p(t_min), _p(t_max),
                                  _p(unused))
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    """-> int64 [n_rays] (render_utils_kernel.cu:38-55,105-121)."""
    rays_d, t_min, t_max = _f32(rays_d), _f32(t_min), _f32(t_max)
    n = torch.empty(t_min.shape[0], dtype=torch.int64)
    lib().esr_oracle_infer_n_samples(_p(rays_d), _p(t_min), _p(t_max), ctypes.c_float(float(stepdist)),
                                     ctypes.c_int64(t_min.shape[0]), _p(n))
    return n


def infer_ray_start_dir(rays_o, rays_d, t_min):
    """-> [rays_start, rays_dir] (render_utils_kernel.cu:58-79)."""
    rays_o, rays_d, t_min = _f32(rays_o), _f32(rays_d), _f32(t_min)
    start, dirs = torch.empty_like(rays_o), torch.empty_like(rays_o)
    lib().esr_oracle_infer_ray_start_dir(_p(rays_o), _p(rays_d), _p(t_min), ctypes.c_int64(rays_o.shape[0]), _p(start), _p(dirs))
    return [start, dirs]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, n_samples):
    """-> [rays_pts [n, S, 3], mask_outbbox [n, S]] (render_utils_kernel.cu:243-292)."""
    rays_o, rays_d, xyz_min, xyz_max = _f32(rays_o), _f32(rays_d), _f32(xyz_min), _f32(xyz_max)
    n, S = rays_o.shape[0], int(n_samples)
    pts = torch.empty(n, S, 3, dtype=torch.float32)
    mask = torch.empty(n, S, dtype=torch.uint8)
    lib().esr_oracle_sample_ndc_pts(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), ctypes.c_int(S), ctypes.c_int64(n),
                                    _p(pts), _p(mask))
    return [pts, mask.bool()]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, n_samples):
    """-> rays_pts [n, S, 3] (render_utils_kernel.cu:294-360)."""
    rays_o, rays_d, t_max = _f32(rays_o), _f32(rays_d), _f32(t_max)
    n, S = rays_o.shape[0], int(n_samples)
    pts = torch.empty(n, S, 3, dtype=torch.float32)
    lib().esr_oracle_sample_bg_pts(_p(rays_o), _p(rays_d), _p(t_max), ctypes.c_float(float(bg_preserve)), ctypes.c_int(S),
                                   ctypes.c_int64(n), _p(pts))
    return pts


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """-> bool [n_pts] (render_utils_kernel.cu:366-423)."""
    w = world.to(torch.uint8).contiguous()
    xyz, sc, sh = _f32(xyz), _f32(xyz2ijk_scale), _f32(xyz2ijk_shift)
    out = torch.zeros(xyz.shape[0], dtype=torch.uint8)
    lib().esr_oracle_maskcache_lookup(_p(w), _p(xyz), _p(sc), _p(sh), ctypes.c_int(w.shape[0]), ctypes.c_int(w.shape[1]),
                                      ctypes.c_int(w.shape[2]), ctypes.c_int64(xyz.shape[0]), _p(out))
    return out.bool()


def _raw2alpha(density, shift, interval, interval_t):
    density = _f32(density)
    e, a = torch.empty_like(density), torch.empty_like(density)
    it = _f32(interval_t) if interval_t is not None else None
    lib().esr_oracle_raw2alpha(_p(density), ctypes.c_float(float(shift)), ctypes.c_float(float(interval)),
                               _p(it) if it is not None else None, ctypes.c_int64(density.shape[0]), _p(e), _p(a))
    return [e, a]


def _raw2alpha_bwd(exp_d, grad_back, interval, interval_t):
    exp_d, grad_back = _f32(exp_d), _f32(grad_back)
    g = torch.empty_like(exp_d)
    it = _f32(interval_t) if interval_t is not None else None
    lib().esr_oracle_raw2alpha_bwd(_p(exp_d), _p(grad_back), ctypes.c_float(float(interval)), _p(it) if it is not None else None,
                                   ctypes.c_int64(exp_d.shape[0]), _p(g))
    return g


def raw2alpha(density, shift, interval):
    """-> [exp, alpha] (render_utils_kernel.cu:431-443,462-482)."""
    return _raw2alpha(density, shift, interval, None)


def raw2alpha_nonuni(density, shift, interval):
    return _raw2alpha(density, shift, 0.0, interval)


def raw2alpha_backward(exp_d, grad_back, interval):
    """-> grad (render_utils_kernel.cu:504-515,532-552)."""
    return _raw2alpha_bwd(exp_d, grad_back, interval, None)


def raw2alpha_nonuni_backward(exp_d, grad_back, interval):
    return _raw2alpha_bwd(exp_d, grad_back, 0.0, interval)


def total_variation_add_grad_new(param, grad, mask, wx, wy, wz, dense_mode):
    """In-place on `grad` (total_variation_kernel.cu:38-66,101-131)."""
    assert param.is_contiguous() and grad.is_contiguous() and mask.is_contiguous()
    assert param.dtype == grad.dtype == mask.dtype == torch.float32
    lib().esr_oracle_tv_add_grad_masked(
        _p(param.detach()), _p(grad), _p(mask),
        ctypes.c_float(float(wx)), ctypes.c_float(float(wy)), ctypes.c_float(float(wz)),
        ctypes.c_int64(param.shape[2]), ctypes.c_int64(param.shape[3]), ctypes.c_int64(param.shape[4]),
        ctypes.c_int64(param.numel()), ctypes.c_int(1 if dense_mode else 0))


def sample_pts_on_rays_fma(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """WHAT-IF variant of the fp32 sampler with every multiply-add fused (esr_oracle.c: what nvcc's default contraction could
    make of the reference's statements) -> [ray_pts, mask_outbbox, N_steps, t_min, t_max].  Measurement only."""
    rays_o, rays_d, xyz_min, xyz_max = _f32(rays_o), _f32(rays_d), _f32(xyz_min), _f32(xyz_max)
    n = rays_o.shape[0]
    t_min, t_max = torch.empty(n, dtype=torch.float32), torch.empty(n, dtype=torch.float32)
    n_steps = torch.empty(n, dtype=torch.int64)
    L = lib()
    L.esr_oracle_sample_count_fma(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), ctypes.c_float(float(near)),
                                  ctypes.c_float(float(far)), ctypes.c_float(float(stepdist)), ctypes.c_int64(n),
                                  _p(t_min), _p(t_max), _p(n_steps))
    total = int(n_steps.sum().item())
    pts, mask = torch.empty(total, 3, dtype=torch.float32), torch.empty(total, dtype=torch.uint8)
    L.esr_oracle_sample_fill_fma(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _p(t_min), _p(n_steps),
                                 ctypes.c_float(float(stepdist)), ctypes.c_int64(n), _p(pts), _p(mask))
    return [pts, mask.bool(), n_steps, t_min, t_max]


class _Namespace:
    """Stand-in for the pybind module object the reference imports."""


def as_render_utils_module():
    m = _Namespace()
    m.sample_pts_on_rays = sample_pts_on_rays
    m.alpha2weight = alpha2weight
    m.alpha2weight_backward = alpha2weight_backward
    for f in (infer_t_minmax, infer_n_samples, infer_ray_start_dir, sample_ndc_pts_on_rays, sample_bg_pts_on_rays,
              maskcache_lookup, raw2alpha, raw2alpha_backward, raw2alpha_nonuni, raw2alpha_nonuni_backward):
        setattr(m, f.__name__, f)               # (exported by the reference's module, never called by its Python)
    return m


def as_total_variation_module():
    m = _Namespace()
    m.total_variation_add_grad = total_variation_add_grad
    m.total_variation_add_grad_new = total_variation_add_grad_new
    return m


__all__ = [
    "build", "sample_pts_on_rays", "alpha2weight", "alpha2weight_backward",
    "total_variation_add_grad", "segment_sum",
    "as_render_utils_module", "as_total_variation_module",
]
_ = np  # numpy kept importable for callers that pass arrays through torch.from_numpy
