"""Drop-in for the reference's two pybind extension modules.

Same function names, argument order and return lists as
``render_utils_cuda`` (app/utils/base/cuda/render_utils.cpp:170-184) and
``total_variation_cuda`` (app/utils/base/cuda/total_variation.cpp:29-32), so the
reference's ``app/utils/base/module.py`` / ``voxurff.py`` call sites run on it
unchanged; the bodies call the C ABI of libesr_hip.so.  Errors surface as
RuntimeError like the reference's TORCH_CHECK(is_cuda / is_contiguous).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_i64 = torch.int64
_f32 = torch.float32


def _chk(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}")


def _float_type(t: torch.Tensor, name: str):
    """The reference dispatches its three live ops over float AND double (AT_DISPATCH_FLOATING_TYPES): fp32 runs the tuned
    kernels, fp64 the literal restatement of the double instantiation (entry points *_f64); anything else is an error."""
    if t.dtype not in (_f32, torch.float64):
        raise RuntimeError(f"{name} must be float32 or float64")
    return t.dtype


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> [ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max]."""
    ft = _float_type(rays_o, "rays_o")
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        _chk(t, n, ft)
    L = _lib.lib()
    count, fill = (L.esr_sample_count, L.esr_sample_fill) if ft == _f32 else (L.esr_sample_count_f64, L.esr_sample_fill_f64)
    dev = rays_o.device
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=ft, device=dev)
    t_max = torch.empty(n, dtype=ft, device=dev)
    n_steps = torch.empty(n, dtype=_i64, device=dev)
    cumsum = torch.empty(n, dtype=_i64, device=dev)
    total = torch.empty(1, dtype=_i64, device=dev)
    s = _lib.stream_ptr(dev)
    _lib.check(count(
        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
        C.c_float(float(near)), C.c_float(float(far)), C.c_float(float(stepdist)), C.c_int64(n),
        _lib.ptr(t_min), _lib.ptr(t_max), _lib.ptr(n_steps), _lib.ptr(cumsum), _lib.ptr(total), s),
        "esr_sample_count")
    m = int(total.item())          # same device->host sync as the reference (kernel.cu:212)
    ray_pts = torch.empty(m, 3, dtype=ft, device=dev)
    mask = torch.empty(m, dtype=torch.bool, device=dev)
    ray_id = torch.empty(m, dtype=_i64, device=dev)
    step_id = torch.empty(m, dtype=_i64, device=dev)
    _lib.check(fill(
        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(xyz_min), _lib.ptr(xyz_max), _lib.ptr(t_min),
        _lib.ptr(cumsum), C.c_float(float(stepdist)), C.c_int64(n), C.c_int64(m),
        _lib.ptr(ray_pts), _lib.ptr(mask), _lib.ptr(ray_id), _lib.ptr(step_id), s),
        "esr_sample_fill")
    return [ray_pts, mask, ray_id, step_id, n_steps, t_min, t_max]


def alpha2weight(alpha, ray_id, n_rays):
    """-> [weight, T, alphainv_last, i_start, i_end]."""
    ft = _float_type(alpha, "alpha")
    _chk(alpha, "alpha", ft)
    _chk(ray_id, "ray_id", _i64)
    L = _lib.lib()
    dev = alpha.device
    m, n_rays = alpha.shape[0], int(n_rays)
    weight = torch.empty_like(alpha)
    T = torch.empty_like(alpha)
    last = torch.empty(n_rays, dtype=ft, device=dev)
    i_s = torch.empty(n_rays, dtype=_i64, device=dev)
    i_e = torch.empty(n_rays, dtype=_i64, device=dev)
    _lib.check((L.esr_alpha2weight_fwd if ft == _f32 else L.esr_alpha2weight_fwd_f64)(
        _lib.ptr(alpha), _lib.ptr(ray_id), C.c_int64(m), C.c_int64(n_rays), _lib.ptr(weight),
        _lib.ptr(T), _lib.ptr(last), _lib.ptr(i_s), _lib.ptr(i_e), _lib.stream_ptr(dev)),
        "esr_alpha2weight_fwd")
    return [weight, T, last, i_s, i_e]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays,
                          grad_weights, grad_last):
    ft = _float_type(alpha, "alpha")
    for t, n in ((alpha, "alpha"), (weight, "weight"), (T, "T"), (alphainv_last, "alphainv_last"),
                 (grad_weights, "grad_weights"), (grad_last, "grad_last")):
        _chk(t, n, ft)
    L = _lib.lib()
    grad = torch.empty_like(alpha)
    _lib.check((L.esr_alpha2weight_bwd if ft == _f32 else L.esr_alpha2weight_bwd_f64)(
        _lib.ptr(alpha), _lib.ptr(weight), _lib.ptr(T), _lib.ptr(alphainv_last), _lib.ptr(i_start),
        _lib.ptr(i_end), C.c_int64(alpha.shape[0]), C.c_int64(int(n_rays)), _lib.ptr(grad_weights),
        _lib.ptr(grad_last), _lib.ptr(grad), _lib.stream_ptr(alpha.device)),
        "esr_alpha2weight_bwd")
    return grad


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """In place on ``grad`` ([1,C,X,Y,Z] like ``param``)."""
    _chk(param, "param", _f32)
    _chk(grad, "grad", _f32)
    L = _lib.lib()
    _lib.check(L.esr_tv_add_grad(
        _lib.ptr(param), _lib.ptr(grad), C.c_float(float(wx)), C.c_float(float(wy)),
        C.c_float(float(wz)), C.c_int64(param.shape[2]), C.c_int64(param.shape[3]),
        C.c_int64(param.shape[4]), C.c_int64(param.numel()), C.c_int(1 if dense_mode else 0),
        _lib.stream_ptr(param.device)), "esr_tv_add_grad")


def segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.segment_coo(src, index, out, reduce='sum') for a sorted index."""
    if reduce != "sum":
        raise NotImplementedError("only reduce='sum' is on the path")
    _chk(index, "index", _i64)
    src = src.contiguous()
    _chk(src, "src", _f32)
    if out is None:
        out = torch.zeros((int(dim_size),) + tuple(src.shape[1:]), dtype=_f32, device=src.device)
    _chk(out, "out", _f32)
    c = 1 if src.dim() == 1 else src.shape[1]
    _lib.check(_lib.lib().esr_segment_sum(
        _lib.ptr(src), _lib.ptr(index), C.c_int64(src.shape[0]), C.c_int64(c), _lib.ptr(out),
        C.c_int64(out.shape[0]), _lib.stream_ptr(src.device)), "esr_segment_sum")
    return out


# ---- exported by the reference's extensions but never called from its Python: same names, argument order
# and return lists (render_utils.cpp:171-173,175-181, total_variation.cpp:31); elementwise launches of csrc/legacy_ops.hip.
# FLOAT32 ONLY: the reference dispatches these over float and double (AT_DISPATCH_FLOATING_TYPES); here a float64 tensor
# raises RuntimeError (what TORCH_CHECK raises) -- only the three ops the reference actually calls (sample_pts_on_rays,
# alpha2weight, alpha2weight_backward) carry its double instantiation (the *_f64 entry points).  INTEGRATION.md says the same.
def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    """-> [t_min, t_max]."""
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        _chk(t, n, _f32)
    n = rays_o.shape[0]
    t_min, t_max = torch.empty(n, dtype=_f32, device=rays_o.device), torch.empty(n, dtype=_f32, device=rays_o.device)
    _lib.check(_lib.lib().esr_infer_t_minmax(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                                             C.c_float(float(near)), C.c_float(float(far)), C.c_int64(n), _lib.ptr(t_min),
                                             _lib.ptr(t_max), _lib.stream_ptr(rays_o.device)), "esr_infer_t_minmax")
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    for t, n in ((rays_d, "rays_d"), (t_min, "t_min"), (t_max, "t_max")):
        _chk(t, n, _f32)
    n = t_min.shape[0]
    out = torch.empty(n, dtype=_i64, device=t_min.device)
    _lib.check(_lib.lib().esr_infer_n_samples(_lib.ptr(rays_d), _lib.ptr(t_min), _lib.ptr(t_max), C.c_float(float(stepdist)),
                                              C.c_int64(n), _lib.ptr(out), _lib.stream_ptr(t_min.device)), "esr_infer_n_samples")
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    """-> [rays_start, rays_dir]."""
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (t_min, "t_min")):
        _chk(t, n, _f32)
    start, dirs = torch.empty_like(rays_o), torch.empty_like(rays_o)
    _lib.check(_lib.lib().esr_infer_ray_start_dir(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_min), C.c_int64(rays_o.shape[0]),
                                                  _lib.ptr(start), _lib.ptr(dirs), _lib.stream_ptr(rays_o.device)),
               "esr_infer_ray_start_dir")
    return [start, dirs]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    """-> [rays_pts [n, N, 3], mask_outbbox [n, N]]."""
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        _chk(t, n, _f32)
    n, S = rays_o.shape[0], int(N_samples)
    pts = torch.empty(n, S, 3, dtype=_f32, device=rays_o.device)
    mask = torch.empty(n, S, dtype=torch.bool, device=rays_o.device)
    _lib.check(_lib.lib().esr_sample_ndc_pts(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                                             C.c_int32(S), C.c_int64(n), _lib.ptr(pts), _lib.ptr(mask),
                                             _lib.stream_ptr(rays_o.device)), "esr_sample_ndc_pts")
    return [pts, mask]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (t_max, "t_max")):
        _chk(t, n, _f32)
    n, S = rays_o.shape[0], int(N_samples)
    pts = torch.empty(n, S, 3, dtype=_f32, device=rays_o.device)
    _lib.check(_lib.lib().esr_sample_bg_pts(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t_max), C.c_float(float(bg_preserve)),
                                            C.c_int32(S), C.c_int64(n), _lib.ptr(pts), _lib.stream_ptr(rays_o.device)),
               "esr_sample_bg_pts")
    return pts


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """world: bool [I, J, K]; xyz [n, 3] -> bool [n] (False outside the volume)."""
    _chk(world, "world", torch.bool)
    for t, n in ((xyz, "xyz"), (xyz2ijk_scale, "xyz2ijk_scale"), (xyz2ijk_shift, "xyz2ijk_shift")):
        _chk(t, n, _f32)
    out = torch.zeros(xyz.shape[0], dtype=torch.bool, device=xyz.device)
    _lib.check(_lib.lib().esr_maskcache_lookup(_lib.ptr(world), _lib.ptr(xyz), _lib.ptr(xyz2ijk_scale), _lib.ptr(xyz2ijk_shift),
                                               C.c_int32(world.shape[0]), C.c_int32(world.shape[1]), C.c_int32(world.shape[2]),
                                               C.c_int64(xyz.shape[0]), _lib.ptr(out), _lib.stream_ptr(xyz.device)),
               "esr_maskcache_lookup")
    return out


def _raw2alpha(density, shift, interval, per_point):
    _chk(density, "density", _f32)
    if per_point is not None:
        _chk(per_point, "interval", _f32)
    e, a = torch.empty_like(density), torch.empty_like(density)
    _lib.check(_lib.lib().esr_raw2alpha(_lib.ptr(density), C.c_float(float(shift)), C.c_float(float(interval)),
                                        _lib.ptr(per_point) if per_point is not None else None, C.c_int64(density.shape[0]),
                                        _lib.ptr(e), _lib.ptr(a), _lib.stream_ptr(density.device)), "esr_raw2alpha")
    return [e, a]


def _raw2alpha_bwd(exp, grad_back, interval, per_point):
    _chk(exp, "exp", _f32)
    _chk(grad_back, "grad_back", _f32)
    if per_point is not None:
        _chk(per_point, "interval", _f32)
    g = torch.empty_like(exp)
    _lib.check(_lib.lib().esr_raw2alpha_bwd(_lib.ptr(exp), _lib.ptr(grad_back), C.c_float(float(interval)),
                                            _lib.ptr(per_point) if per_point is not None else None, C.c_int64(exp.shape[0]),
                                            _lib.ptr(g), _lib.stream_ptr(exp.device)), "esr_raw2alpha_bwd")
    return g


def raw2alpha(density, shift, interval):
    """-> [exp, alpha]."""
    return _raw2alpha(density, shift, interval, None)


def raw2alpha_nonuni(density, shift, interval):
    """``interval``: a tensor, one value per point."""
    return _raw2alpha(density, shift, 0.0, interval)


def raw2alpha_backward(exp, grad_back, interval):
    return _raw2alpha_bwd(exp, grad_back, interval, None)


def raw2alpha_nonuni_backward(exp, grad_back, interval):
    return _raw2alpha_bwd(exp, grad_back, 0.0, interval)


def total_variation_add_grad_new(param, grad, mask, wx, wy, wz, dense_mode):
    """In place on ``grad``; ``mask``: a float tensor shaped like ``param``."""
    for t, n in ((param, "param"), (grad, "grad"), (mask, "mask")):
        _chk(t, n, _f32)
    _lib.check(_lib.lib().esr_tv_add_grad_masked(
        _lib.ptr(param), _lib.ptr(grad), _lib.ptr(mask), C.c_float(float(wx)), C.c_float(float(wy)), C.c_float(float(wz)),
        C.c_int64(param.shape[2]), C.c_int64(param.shape[3]), C.c_int64(param.shape[4]), C.c_int64(param.numel()),
        C.c_int(1 if dense_mode else 0), _lib.stream_ptr(param.device)), "esr_tv_add_grad_masked")
