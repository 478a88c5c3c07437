"""ctypes binding of libesr_hip.so (include/esr_hip.h).

The library is the product: if it is missing or fails to load, everything that
needs it raises -- there is no CPU or PyTorch fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libesr_hip.so")
# developer experiments only (tools/variant.sh builds alternative libraries; tools/ab_env.sh times them side by side on one box)
LIB_PATH = os.environ.get("ESR_LIB_PATH", LIB_PATH)
ABI_VERSION = 25
_lib = None


class EsrScene(C.Structure):
    _fields_ = [
        ("xyz_min", C.c_float * 3), ("xyz_max", C.c_float * 3),
        ("mask_min", C.c_float * 3), ("mask_max", C.c_float * 3),
        ("gx", C.c_int32), ("gy", C.c_int32), ("gz", C.c_int32),
        ("mx", C.c_int32), ("my", C.c_int32), ("mz", C.c_int32),
        ("near_", C.c_float), ("stepdist", C.c_float), ("voxel_size", C.c_float),
        ("act_shift", C.c_float), ("mask_thres", C.c_float), ("fast_thres", C.c_float),
        ("s_val", C.c_float), ("max_steps", C.c_int32), ("grad_feat", C.c_float * 4),
    ]


class EsrPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_on", "n_off", "tiles_on", "tiles_all", "m0", "m1", "m2", "overflow")]


class EsrFeatArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rays_o", "rays_d", "viewdirs", "rec_ray", "rec_step", "rec_sdf",
                                          "pts", "pt_viewdirs", "pt_sdf")] + \
               [("n_pts", C.c_int32), ("sdf", C.c_void_p), ("color_on", C.c_void_p * 3),
                ("color_off", C.c_void_p * 3), ("tiles_on", C.c_int32), ("tiles_all", C.c_int32)]


class EsrFeatBwdSrc(C.Structure):
    _fields_ = [("dX", C.c_void_p), ("grad_color_on", C.c_void_p), ("grad_color_off", C.c_void_p),
                ("t0", C.c_int32), ("t1", C.c_int32)]


class EsrLtsArgs(C.Structure):
    _fields_ = [("n_pts", C.c_int32), ("n_rays", C.c_int32), ("n_sg", C.c_int32), ("pdra_mode", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("base", "rough", "metal", "normal", "view", "dirs", "off_m", "emo_m",
                                          "last2", "mus", "lambdas", "lobes", "emission", "umask")]


class EsrLtsGather(C.Structure):          # esr_lts_gather_t
    _fields_ = [(n, C.c_void_p) for n in ("jp", "ray64", "pts_all", "eg", "rec_sdf", "viewdirs", "brdf_a", "emit_a",
                                          "umask_rays")] + [("n_pts", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("pts2", "vd2", "sdf2", "normal", "base", "rough", "metal", "emis", "umask", "pt1")]


class EsrActJob(C.Structure):             # esr_act_job_t
    _fields_ = [("z", C.c_void_p), ("g_tile", C.c_void_p), ("out", C.c_void_p), ("tiles", C.c_int32), ("rows", C.c_int32),
                ("n_ch", C.c_int32), ("act", C.c_int32), ("bwd", C.c_int32), ("src", C.c_void_p), ("src_c", C.c_int32),
                ("n_src", C.c_int32), ("inv", C.c_void_p), ("pt1", C.c_void_p), ("ex", C.c_void_p * 3),
                ("ex_c", C.c_int32 * 3), ("ex_col0", C.c_int32 * 3)]


class EsrGatherJob(C.Structure):          # esr_gather_job_t
    _fields_ = [("src", C.c_void_p), ("tile_rows", C.c_int32), ("row_stride", C.c_int32), ("col0", C.c_int32),
                ("n_ch", C.c_int32), ("perm", C.c_void_p), ("n", C.c_int32), ("out", C.c_void_p)]


class EsrPairJob(C.Structure):            # esr_pair_job_t
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int32), ("row_mask", C.c_void_p),
                ("mask_value", C.c_int32), ("count_dev", C.c_void_p), ("kind", C.c_int32), ("w_value", C.c_float),
                ("w_a", C.c_float), ("w_b", C.c_float), ("ga", C.c_void_p), ("gb", C.c_void_p)]


class EsrLtsGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_off_m", "d_emo_m", "d_last2", "d_base", "d_rough", "d_metal",
                                          "d_emission", "d_mus", "d_lambdas", "d_lobes")]


class EsrMlpWeights(C.Structure):
    _fields_ = [("w", C.c_void_p * 4), ("b", C.c_void_p * 4)]


class EsrWgradJob(C.Structure):
    """esr_wgrad_job_t"""
    _fields_ = [("kind", C.c_int32), ("color_row0", C.c_int32), ("t0", C.c_int32), ("t1", C.c_int32),
                ("X", C.c_void_p), ("H", C.c_void_p), ("dZ", C.c_void_p), ("dz", C.c_void_p),
                ("gw", C.c_void_p), ("gb", C.c_void_p), ("X16", C.c_void_p), ("amax", C.c_void_p)]


# every exported symbol of include/esr_hip.h (checked by tests/test_host.py::test_library_loads_and_exports_every_header_symbol)
EXPORTS = [
    "esr_abi_version", "esr_build_info",
    "esr_sample_count", "esr_sample_fill", "esr_alpha2weight_fwd", "esr_alpha2weight_bwd",
    "esr_tv_add_grad", "esr_segment_sum",
    "esr_sample_count_f64", "esr_sample_fill_f64", "esr_alpha2weight_fwd_f64", "esr_alpha2weight_bwd_f64",
    "esr_infer_t_minmax", "esr_infer_n_samples", "esr_infer_ray_start_dir", "esr_sample_ndc_pts", "esr_sample_bg_pts",
    "esr_maskcache_lookup", "esr_raw2alpha", "esr_raw2alpha_bwd", "esr_tv_add_grad_masked",
    "esr_fine_march_count", "esr_fine_plan_begin", "esr_fine_plan", "esr_fine_plan_totals", "esr_fine_plan_offsets", "esr_fine_march_fill",
    "esr_fine_march_bwd", "esr_fine_march_bwd_rec", "esr_fine_march_cache_floats", "esr_fine_march_count_cached",
    "esr_fine_march_fill_cached", "esr_fine_march_bwd_cached", "esr_fine_march_count_ga", "esr_fine_march_fill_ga", "esr_fine_march_bwd_ga",
    "esr_fine_feat_fwd", "esr_fine_feat_fwd_x16", "esr_fine_feat_x16_bytes", "esr_fine_feat_bwd",
    "esr_mlp_packed_floats", "esr_mlp_pack", "esr_mlp_pack_batch", "esr_mlp_packed_split_elems", "esr_mlp_split_gain_offset", "esr_mlp_split_range_flag", "esr_mlp_fwd_split", "esr_mlp_fwd_fine_split", "esr_mlp_dgrad_split", "esr_mlp_dgrad_fine_split", "esr_absmax", "esr_mlp_fwd", "esr_mlp_fwd_mixed", "esr_mlp_fwd_fine", "esr_mlp_dgrad_fine", "esr_mlp_fwd_fine_bf16", "esr_mlp_dgrad_fine_bf16", "esr_mlp_dgrad", "esr_mlp_dgrad_wg", "esr_mlp_wgrad", "esr_mlp_wgrad_batch", "esr_tone_wgrad_scratch_floats", "esr_tone_wgrad_recompute", "esr_tone_wgrad_recompute_bf16", "esr_tone_wgrad_recompute_split",
    "esr_mlp_wgrad_scratch_floats",
    "esr_fine_tone_in_fwd", "esr_fine_composite_fwd", "esr_fine_composite_bwd",
    "esr_fine_tone_in_bwd", "esr_fine_loss_fwd_bwd", "esr_fine_loss_fwd_bwd_dp",
    "esr_expgrad_fwd", "esr_expgrad_bwd", "esr_lts_dirs", "esr_lts_ref_order", "esr_lts_perturb", "esr_lts_gather_rows",
    "esr_lts_gather_points", "esr_lts_combine_fwd", "esr_lts_combine_bwd",
    "esr_act_fwd", "esr_act_bwd", "esr_act_batch", "esr_lts_gather_rows_batch", "esr_pair_loss_batch", "esr_lts_ref_order_inv", "esr_lts_dirs_rays", "esr_composite3_fwd", "esr_composite3_bwd", "esr_lts_tone_in_bwd",
    "esr_sample_points", "esr_pair_loss_fwd_bwd", "esr_emit_edit",
    "esr_gauss3d_fwd", "esr_gauss3d_bwd", "esr_central_grad_fwd", "esr_central_grad_bwd",
    "esr_coarse_march_count", "esr_coarse_march_fill", "esr_coarse_march_bwd", "esr_coarse_march_count_ga", "esr_coarse_march_fill_ga", "esr_coarse_march_bwd_ga",
    "esr_coarse_feat_fwd", "esr_coarse_feat_bwd", "esr_coarse_shade_fwd", "esr_coarse_shade_bwd",
    "esr_adam_step", "esr_eval_aux", "esr_eval_disp",
    "esr_mlp_packed_bf16_elems", "esr_mlp_pack_bf16", "esr_mlp_fwd_bf16", "esr_mlp_dgrad_bf16", "esr_mlp_wgrad_bf16",
    "esr_brick_floats", "esr_brick_flags", "esr_brick_pack", "esr_brick_unpack", "esr_brick_list", "esr_brick_list_scratch_ints",
    "esr_smooth_grad_tv_fwd", "esr_smooth_grad_tv_bwd", "esr_host_choice_noreplace", "esr_host_choice_start", "esr_host_choice_wait",
]


def lib() -> C.CDLL:
    """Load the HIP library; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m esr_nerf_amd.build` "
                "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH)
        L.esr_abi_version.restype = C.c_int
        L.esr_build_info.restype = C.c_char_p
        if hasattr(L, "esr_mlp_packed_floats"):
            L.esr_mlp_packed_floats.restype = C.c_int64
            L.esr_mlp_wgrad_scratch_floats.restype = C.c_int64
            L.esr_tone_wgrad_scratch_floats.restype = C.c_int64
            L.esr_fine_march_cache_floats.restype = C.c_int64
        if hasattr(L, "esr_fine_feat_x16_bytes"):
            L.esr_fine_feat_x16_bytes.restype = C.c_int64
        if hasattr(L, "esr_brick_list_scratch_ints"):
            L.esr_brick_list_scratch_ints.restype = C.c_int64
        if hasattr(L, "esr_mlp_packed_bf16_elems"):
            L.esr_mlp_packed_bf16_elems.restype = C.c_int64
        if hasattr(L, "esr_mlp_packed_split_elems"):
            L.esr_mlp_packed_split_elems.restype = C.c_int64
        if hasattr(L, "esr_mlp_split_gain_offset"):
            L.esr_mlp_split_gain_offset.restype = C.c_int64
        if L.esr_abi_version() != ABI_VERSION:
            raise RuntimeError("libesr_hip.so ABI version mismatch: rebuild")
        _lib = L
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}" +
                           (" (hipError)" if code > 0 else
                            " (a capacity limit of the kernels is exceeded, include/esr_hip.h: ESR_ECAP)" if code == -2 else
                            " (bad argument)"))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor (or NULL for None)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("esr_hip ops need device tensors (got a CPU tensor)")
    if not t.is_contiguous():
        raise RuntimeError("esr_hip ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


_dev_index = {}


def stream_ptr(device=None):
    """The current HIP stream of ``device`` as a void pointer.  (``torch.cuda.current_stream(device).cuda_stream`` builds a
    Stream object per call, ~5 us; a light-transport step asks ~100 times -- the raw getter is what torch's own
    compiled code uses.)"""
    idx = _dev_index.get(device)
    if idx is None:
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()            # ("cuda" without an index: not cached)
        elif device is not None:
            _dev_index[device] = idx
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def ptr_array(tensors, n=None):
    n = n or len(tensors)
    arr = (C.c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = 0 if t is None else t.data_ptr()
    return arr
