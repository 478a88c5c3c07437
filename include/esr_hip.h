/*
 * esr_hip.h -- C ABI of libesr_hip.so, the MI355X (gfx950) implementation of
 * ESR-NeRF's volumetric-rendering hot path.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is CALLER-OWNED DEVICE memory, contiguous, unless marked
 *     [host]; outputs are never allocated inside the library.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     kernel is enqueued on it, nothing synchronises the device.
 *   - return value: 0 on success, otherwise the (positive) hipError_t of the
 *     failing runtime call, or a negative ESR_E* argument error.  Nothing throws
 *     across the ABI.  The Python shim re-raises non-zero codes as RuntimeError,
 *     which is what the reference's TORCH_CHECK failures surface as
 *     (app/utils/base/cuda/render_utils.cpp:46-48).
 *   - re-entrant.  The only process-wide state is an idempotent per-(kernel, device) memo of the > 64 KB
 *     dynamic-LDS opt-in (hipFuncSetAttribute); no environment variable is read on a launch path.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   app/utils/base/cuda/render_utils.cpp:170-184  pybind module render_utils_cuda
 *   app/utils/base/cuda/total_variation.cpp:29-32 pybind module total_variation_cuda
 *   torch_scatter.segment_coo (third party; call sites app/fine/model/voxurff.py:260-272)
 *   and, for the fused fine-stage ops, the torch-level body of
 *   VoxurfF.forward_training (app/fine/model/voxurff.py:177-278).
 */
#ifndef ESR_HIP_H
#define ESR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESR_EINVAL (-1)   /* bad argument (null pointer, negative size, bad dims) */
#define ESR_ECAP   (-2)   /* a capacity limit of the kernels is exceeded          */

#define ESR_TILE 32       /* samples per tile of the tile-major activation layout  */

int esr_abi_version(void);          /* bumps whenever a signature below changes */
const char *esr_build_info(void);   /* "gfx950 <date>"                         */

/* ------------------------------------------------------------------------- *
 * A. Drop-in replacements of the reference's native ops (same arithmetic,
 *    same outputs, caller allocates).
 * ------------------------------------------------------------------------- */

/*
 * sample_pts_on_rays, phase 1 -- replaces infer_t_minmax + infer_n_samples +
 * N_steps.cumsum + N_steps.sum (render_utils_kernel.cu:12-55,204-212).
 * rays_o, rays_d [n_rays,3] f32; xyz_min, xyz_max [3] f32 (device, as in the
 * reference).  Outputs: t_min, t_max [n_rays] f32; n_steps, cumsum [n_rays] i64
 * (inclusive scan); total [1] i64 (device).  The caller reads `total` back to
 * size phase 2 -- the same device->host sync the reference performs at
 * render_utils_kernel.cu:212.
 */
int esr_sample_count(const float *rays_o, const float *rays_d, const float *xyz_min,
                     const float *xyz_max, float near_, float far_, float stepdist,
                     int64_t n_rays, float *t_min, float *t_max, int64_t *n_steps,
                     int64_t *cumsum, int64_t *total, void *stream);

/*
 * sample_pts_on_rays, phase 2 -- replaces __set_1_at_ray_seg_start + cumsum_ +
 * __set_step_id + infer_ray_start_dir + sample_pts_on_rays_cuda_kernel
 * (render_utils_kernel.cu:58-79,144-194,213-241).  Outputs have `total` rows:
 * ray_pts [total,3] f32, mask_outbbox [total] u8 (bool), ray_id, step_id [total] i64.
 */
int esr_sample_fill(const float *rays_o, const float *rays_d, const float *xyz_min,
                    const float *xyz_max, const float *t_min, const int64_t *cumsum,
                    float stepdist, int64_t n_rays, int64_t total, float *ray_pts,
                    uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id, void *stream);

/*
 * alpha2weight -- replaces render_utils_kernel.cu:577-651.  alpha [n_pts] f32,
 * ray_id [n_pts] i64 sorted.  Writes weight, T [n_pts]; alphainv_last [n_rays];
 * i_start, i_end [n_rays] i64 -- all fully initialised here (0/1/1/0/0 defaults).
 */
int esr_alpha2weight_fwd(const float *alpha, const int64_t *ray_id, int64_t n_pts,
                         int64_t n_rays, float *weight, float *T, float *alphainv_last,
                         int64_t *i_start, int64_t *i_end, void *stream);

/* alpha2weight_backward -- replaces render_utils_kernel.cu:654-707. grad [n_pts]. */
int esr_alpha2weight_bwd(const float *alpha, const float *weight, const float *T,
                         const float *alphainv_last, const int64_t *i_start,
                         const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                         const float *grad_weights, const float *grad_last, float *grad,
                         void *stream);

/*
 * total_variation_add_grad -- replaces total_variation_kernel.cu:13-35,68-98.
 * In place on grad; param/grad have n elements laid out [..., sz_i, sz_j, sz_k].
 * Keeps the reference's quirk: the i and k axes both use wz, wx is unused.
 */
int esr_tv_add_grad(const float *param, float *grad, float wx, float wy, float wz,
                    int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n, int dense_mode,
                    void *stream);

/*
 * Sorted segment sum -- replaces torch_scatter.segment_coo(src, index, out,
 * reduce="sum") (voxurff.py:260-272).  out [n_seg,c] is accumulated into
 * (caller zero-fills); index [n] i64 non-decreasing.
 */
int esr_segment_sum(const float *src, const int64_t *index, int64_t n, int64_t c,
                    float *out, int64_t n_seg, void *stream);

/*
 * The reference's DOUBLE instantiation of its three live ops (AT_DISPATCH_FLOATING_TYPES, render_utils_kernel.cu:229,639,692;
 * nothing in the reference's Python produces double tensors, the shim dispatches on dtype like the reference does).  Same
 * contracts as the fp32 entry points with double tensors.  The reference's kernels keep `float` locals whatever the tensor
 * type: t_min / t_max, the sample points, the running transmittance and the backward's running sum are rounded to float on
 * the way (stored as doubles) -- restated literally, one thread per ray / sample (csrc/sampler.hip, composite.hip).
 */
int esr_sample_count_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                         float near_, float far_, float stepdist, int64_t n_rays, double *t_min, double *t_max,
                         int64_t *n_steps, int64_t *cumsum, int64_t *total, void *stream);
int esr_sample_fill_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                        const double *t_min, const int64_t *cumsum, float stepdist, int64_t n_rays, int64_t total,
                        double *ray_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id, void *stream);
int esr_alpha2weight_fwd_f64(const double *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays, double *weight,
                             double *T, double *alphainv_last, int64_t *i_start, int64_t *i_end, void *stream);
int esr_alpha2weight_bwd_f64(const double *alpha, const double *weight, const double *T, const double *alphainv_last,
                             const int64_t *i_start, const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                             const double *grad_weights, const double *grad_last, double *grad, void *stream);

/*
 * The ops the two pybind modules EXPORT but the reference's own Python never calls (render_utils.cpp:171-173,175-181,
 * total_variation.cpp:31).  Not on the accelerated path: one thread per ray / point / cell, the fp32
 * instantiation of the reference's templates statement by statement (csrc/legacy_ops.hip; float / double mixing where the
 * reference's literals put it).  Caller-allocated outputs; bool tensors are bytes.
 */
/* infer_t_minmax -- render_utils_kernel.cu:12-35,82-103: ray / box range clamped into [near, far] -> t_min, t_max [n_rays]. */
int esr_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                       float near_, float far_, int64_t n_rays, float *t_min, float *t_max, void *stream);
/* infer_n_samples -- :38-55,105-121: max(ceil((t_max - t_min) |d| / stepdist), 1) -> int64 [n_rays]. */
int esr_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max, float stepdist, int64_t n_rays,
                        int64_t *n_samples, void *stream);
/* infer_ray_start_dir -- :58-79,123-140: o + d t_min and d / |d| -> [n_rays,3] each. */
int esr_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min, int64_t n_rays,
                            float *rays_start, float *rays_dir, void *stream);
/* sample_ndc_pts_on_rays -- :243-292: o + d step / (N - 1) -> rays_pts [n_rays,N,3], mask_outbbox [n_rays,N]. */
int esr_sample_ndc_pts(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                       int32_t n_samples, int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox, void *stream);
/* sample_bg_pts_on_rays -- :294-360: inverted-sphere background points -> rays_pts [n_rays,N,3]. */
int esr_sample_bg_pts(const float *rays_o, const float *rays_d, const float *t_max, float bg_preserve, int32_t n_samples,
                      int64_t n_rays, float *rays_pts, void *stream);
/* maskcache_lookup -- :366-423: nearest voxel of a bool volume [sz_i,sz_j,sz_k]; outside reads 0 -> out [n_pts]. */
int esr_maskcache_lookup(const uint8_t *world, const float *xyz, const float *xyz2ijk_scale, const float *xyz2ijk_shift,
                         int32_t sz_i, int32_t sz_j, int32_t sz_k, int64_t n_pts, uint8_t *out, void *stream);
/* raw2alpha / raw2alpha_nonuni -- :431-502: e = exp(density + shift), alpha = 1 - (1 + e)^(-interval) -> exp_d, alpha [n_pts].
 * interval_per_point != NULL: the _nonuni form (one interval per point; `interval` is ignored). */
int esr_raw2alpha(const float *density, float shift, float interval, const float *interval_per_point, int64_t n_pts,
                  float *exp_d, float *alpha, void *stream);
/* raw2alpha_backward / raw2alpha_nonuni_backward -- :504-575: min(e, 1e10) (1 + e)^(-interval - 1) interval grad_back. */
int esr_raw2alpha_bwd(const float *exp_d, const float *grad_back, float interval, const float *interval_per_point,
                      int64_t n_pts, float *grad, void *stream);
/* total_variation_add_grad_new -- total_variation_kernel.cu:38-66,101-131: every term x mask[cell] mask[neighbour] (mask is a
 * float tensor like param); wx on the fastest axis, wy, wz on the slowest (the live kernel's wz-twice quirk is NOT in this one). */
int esr_tv_add_grad_masked(const float *param, float *grad, const float *mask, float wx, float wy, float wz,
                           int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n, int dense_mode, void *stream);

/* ------------------------------------------------------------------------- *
 * B. Fused fine-stage path (VoxurfF.forward_training and its backward).
 * ------------------------------------------------------------------------- */

/* Scene constants, host struct passed by pointer [host] and copied by value. */
typedef struct esr_scene {
    float xyz_min[3], xyz_max[3];     /* sampler box == sdf/colour grid box          */
    float mask_min[3], mask_max[3];   /* mask-cache box                              */
    int32_t gx, gy, gz;               /* sdf / colour grid nodes [X,Y,Z]             */
    int32_t mx, my, mz;               /* mask-cache grid nodes                       */
    float near_, stepdist, voxel_size;
    float act_shift, mask_thres;      /* mask cache: 1-exp(-softplus(d+shift)) >= thres */
    float fast_thres;                 /* alpha > thres, then weight > thres          */
    float s_val;
    int32_t max_steps;                /* upper bound of per-ray step count (LDS sizing) */
    float grad_feat[4];               /* stencil radii of the 24-tap SDF feature (voxels) */
} esr_scene_t;

/*
 * Header the planner leaves in device memory (and the caller reads back once per
 * iteration to size the activation workspace).
 */
typedef struct esr_plan {
    int32_t n_on;        /* surviving samples on emissive-on rays   */
    int32_t n_off;       /* surviving samples on emissive-off rays  */
    int32_t tiles_on;    /* ceil(n_on / 32)                         */
    int32_t tiles_all;   /* tiles_on + ceil(n_off / 32)             */
    int32_t m0, m1, m2;  /* survivors after in-box / mask-cache / alpha>thres */
    int32_t overflow;    /* bit 0: a ray exceeded scene.max_steps; bit 1 (esr_fine_plan): the registered range flag of the
                            split-fp16 forward kernels is set (esr_mlp_split_range_flag) */
} esr_plan_t;

/*
 * march, counting pass: per ray (one wavefront each) sampler -> mask cache ->
 * SDF tap -> NeuS-interp alpha -> alpha>thres -> transmittance with the early
 * stop -> weight>thres (voxurff.py:186-213, functions.py:72-105,
 * render_utils_kernel.cu:577-605).  mask_density [mx,my,mz] (max-pooled),
 * sdf [gx,gy,gz].  Writes cnt3 [n_rays] i32, alphainv_last [n_rays] f32 and ray_stats
 * [n_rays,3] i32 (in-box / mask-cache / alpha>thres survivors of the ray; esr_fine_plan sums
 * them into m0/m1/m2); sets plan->overflow (plan zeroed by esr_fine_plan_begin).
 */
int esr_fine_march_count(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                         const float *mask_density, const float *sdf, int32_t n_rays,
                         int32_t *cnt3, float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan,
                         void *stream);

int esr_fine_plan_begin(esr_plan_t *plan, void *stream);

/*
 * plan: exclusive offsets of every ray's survivors in the compact sample list.
 * Emissive-on rays (em_modes==1) first, then -- starting at the next multiple of
 * 32 -- the off rays, both in ray order, so every 32-sample tile is on-only or
 * off-only.  off3 [n_rays] i32; *plan (device) gets the totals.
 */
int esr_fine_plan(const int32_t *cnt3, const int64_t *em_modes, const int32_t *ray_stats,
                  int32_t n_rays, int32_t *off3, esr_plan_t *plan, void *stream);
/*
 * The same in two launches, for a caller that reads the header back: _totals leaves n_on, n_off, m0, m1, m2 and the overflow
 * word in the header (many workgroups; tiles_on / tiles_all are NOT written: ceil(n_on / 32), tiles_on + ceil(n_off / 32)),
 * _offsets computes off3 (and writes all four counts).  Copy the header between the two: the one-workgroup scan of _offsets
 * then runs while the host reads it.  Header zeroed by esr_fine_plan_begin as for esr_fine_plan.
 */
int esr_fine_plan_totals(const int32_t *cnt3, const int64_t *em_modes, const int32_t *ray_stats, int32_t n_rays,
                         esr_plan_t *plan, void *stream);
int esr_fine_plan_offsets(const int32_t *cnt3, const int64_t *em_modes, int32_t n_rays, int32_t *off3, esr_plan_t *plan,
                          void *stream);

/*
 * march, fill pass: recomputes the march and writes one record per surviving
 * sample at off3[ray] + rank: rec_ray, rec_step (i32), rec_w, rec_sdf (f32).
 * Padding entries must have been set to rec_ray = -1 by the caller.
 */
int esr_fine_march_fill(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                        const float *mask_density, const float *sdf, int32_t n_rays,
                        const int32_t *off3, int32_t *rec_ray, int32_t *rec_step,
                        float *rec_w, float *rec_sdf, void *stream);

/*
 * march backward: d(weights), d(alphainv_last) -> atomic scatter into grad_sdf
 * [gx,gy,gz] through the compositing scan, the alpha formula and the trilinear
 * tap.  dweight [n_tiles*32] is indexed like the records.
 */
int esr_fine_march_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                       const float *mask_density, const float *sdf, int32_t n_rays,
                       const int32_t *off3, const float *dweight, const float *dlast,
                       float *grad_sdf, void *stream);

/*
 * As esr_fine_march_bwd, but the value-tap gradient of every RECORDED sample (the ones esr_fine_march_fill wrote) is
 * added to dsdf_rec [n_tiles*32] (indexed like the records) instead of being scattered: pass that array as
 * `dsdf_extra` to esr_fine_feat_bwd, which folds it into the SDF window it builds anyway (no L2 atomics for it).
 * Samples that were walked but not recorded (below a threshold, yet neighbours of a recorded one) are still scattered
 * into grad_sdf.  accumulate = 0: every recorded slot of dsdf_rec is overwritten (padding slots are left alone; the
 * feature backward never reads them); accumulate = 1: added to what the caller put there.
 */
int esr_fine_march_bwd_rec(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                           const float *mask_density, const float *sdf, int32_t n_rays,
                           const int32_t *off3, const float *dweight, const float *dlast,
                           float *grad_sdf, float *dsdf_rec, int32_t accumulate, void *stream);

/*
 * The three passes with a shared CACHE: the count pass records, for every mask-cache survivor of every ray, its SDF,
 * step id, alpha, transmittance and survivor flags (cache: esr_fine_march_cache_floats(scene, n_rays) floats); the fill
 * pass is then a plain copy of the recorded samples and the backward starts at its reverse scan -- the walk (mask-cache
 * and SDF fetch per step) and the serial transmittance loop run once per step instead of three times.  Results are
 * bit-identical to the uncached entry points.  ray_stats / alphainv_last: the count pass's outputs, untouched since.
 */
int64_t esr_fine_march_cache_floats(const esr_scene_t *scene, int32_t n_rays);
int esr_fine_march_count_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                const float *mask_density, const float *sdf, int32_t n_rays, int32_t *cnt3,
                                float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan, float *cache, void *stream);
int esr_fine_march_fill_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d, int32_t n_rays,
                               const int32_t *off3, const int32_t *ray_stats, const float *cache,
                               int32_t *rec_ray, int32_t *rec_step, float *rec_w, float *rec_sdf, void *stream);
int esr_fine_march_bwd_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d, int32_t n_rays,
                              const int32_t *off3, const int32_t *ray_stats, const float *alphainv_last,
                              const float *cache, const float *dweight, const float *dlast, float *grad_sdf,
                              float *dsdf_rec, int32_t accumulate, void *stream);

/*
 * The same three march entry points for cfg `neus_alpha: grad` (app/utils/base/functions.py:45-69): the section
 * SDFs of a sample are sdf -+ 0.5 * dist * (viewdirs[ray] . grad) with grad = the radius-1 clamped central
 * differences of sample_sdf_grad (app/fine/model/voxurff.py:670-721); `viewdirs` [n_rays,3] is the batch's
 * view-direction tensor.  The backward scatters through the value tap and the six gradient taps.
 */
int esr_fine_march_count_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                            const float *mask_density, const float *sdf, int32_t n_rays, int32_t *cnt3,
                            float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan, void *stream);
int esr_fine_march_fill_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                           const float *mask_density, const float *sdf, int32_t n_rays, const int32_t *off3,
                           int32_t *rec_ray, int32_t *rec_step, float *rec_w, float *rec_sdf, void *stream);
int esr_fine_march_bwd_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                          const float *mask_density, const float *sdf, int32_t n_rays, const int32_t *off3,
                          const float *dweight, const float *dlast, float *grad_sdf, void *stream);

/*
 * Per-sample feature assembly (voxurff.py:219-254 + :678-721 + module.py:24-35; the LTS renderer's
 * esrnerf.py:728-765 uses the same features with three colour grids).
 * Colour grids are channel-last [gx,gy,gz,6] (torch.channels_last_3d storage of the reference's
 * [1,6,X,Y,Z] parameter).  X [n_tiles,104,32] f32, rows:
 *   0-5 colour group 0 | 6 sdf | 7-30 feat24 | 31-42 normal12 | 43-45 xyz | 46-60 sin | 61-75 cos
 *   | 76-84 viewdir PE | 85-87 zero | 88-93 colour group 1 | 94-95 zero | 96-101 colour group 2 | zero
 * gnorm [n_tiles,4,32]: |grad| per stencil radius, kept for the backward.
 */
typedef struct esr_feat_args {
    /* sample positions: either the march records ... */
    const float *rays_o, *rays_d, *viewdirs;        /* [N,3] each; viewdirs per RAY                 */
    const int32_t *rec_ray, *rec_step;              /* [tiles*32]                                   */
    const float *rec_sdf;
    /* ... or explicit points (pts != NULL): per-sample arrays, n_pts valid samples, the rest padding */
    const float *pts, *pt_viewdirs, *pt_sdf;        /* [n_pts,3] [n_pts,3] [n_pts]                  */
    int32_t n_pts;
    const float *sdf;                               /* SDF grid [gx,gy,gz]                          */
    /* grid feeding colour group g on emissive-on tiles / on the other tiles (NULL: zeros).
       fine stage: group0 = {emo, off}, group1 = {off, NULL}; LTS stage: {off,off} {emo,emo} {brdf,brdf} */
    const float *color_on[3], *color_off[3];
    int32_t tiles_on, tiles_all;
} esr_feat_args_t;

int esr_fine_feat_fwd(const esr_scene_t *scene, const esr_feat_args_t *args, float *X, float *gnorm,
                      void *stream);
/*
 * The bf16 engine's form: the input tile goes to X16 as bf16 in the row-quad layout of that engine's saved tiles
 * (esr_fine_feat_x16_bytes(n_tiles) bytes: 26 quads of 256 B per tile = rows 0..95 + the first eight rows once more with
 * the colour group of rows 88..93 in rows 0..5) -- the fp32 rows rounded to bf16, i.e. what the bf16 kernels made of
 * them on load.  X still receives the normal rows 31..42 (the backward reads them), nothing else.  Consumers:
 * esr_mlp_fwd_fine_bf16 (X16 argument; then X may be NULL) and esr_wgrad_job_t::X16.  Stencil radii outside [0, 2] voxels:
 * ESR_ECAP (use esr_fine_feat_fwd).
 */
int64_t esr_fine_feat_x16_bytes(int32_t n_tiles);
int esr_fine_feat_fwd_x16(const esr_scene_t *scene, const esr_feat_args_t *args, float *X, float *gnorm, void *X16,
                          void *stream);

/*
 * Backward.  Every net that consumed the tiles contributes a dX [n_tiles,64,32] (rows 0-42 used)
 * over its tile range: rows 6-42 (sdf value, stencil features, normals) are summed over the sources,
 * rows 0-5 go to that source's colour grid.  dsdf_extra [tiles*32] (optional) is added to the
 * SDF-value row.  With explicit points the SDF-value gradient is returned in dsdf_out [tiles*32]
 * instead of being scattered.  grad_sdf NULL: the SDF grid is frozen, only the colour grids receive
 * gradients (re-lighting fine-tune).  Scatter = LDS accumulation window + z-contiguous float atomics.
 * grad4 [tiles*32][4] (optional; needs grad_sdf): per sample the gradient w.r.t. esr_expgrad_fwd's four outputs at the
 * SAME position (value, d sdf / d x, y, z) -- scattered with esr_expgrad_bwd's weights inside this launch.  grad4_mode:
 * bit 0 = out-of-grid corners dropped (esr_expgrad_bwd's zero_pad) instead of border-replicated; bit 1 = explicit points:
 * the SDF-value gradient (what dsdf_out receives) is scattered the same way, as the value component.
 * The SDF scatter is built for stencil radii (scene->grad_feat) in [0, 2] voxels, the reference's configuration
 * (fine.yaml: [0.5, 1.0, 1.5, 2.0]); other radii return ESR_ECAP (esr_fine_feat_fwd takes any radius).
 */
typedef struct esr_feat_bwd_src {
    const float *dX;
    float *grad_color_on, *grad_color_off;          /* colour grid gradient on on-tiles / other tiles (NULL: none) */
    int32_t t0, t1;
} esr_feat_bwd_src_t;

int esr_fine_feat_bwd(const esr_scene_t *scene, const esr_feat_args_t *args, const float *X,
                      const float *gnorm, const esr_feat_bwd_src_t *src, int32_t n_src,
                      const float *dsdf_extra, float *grad_sdf, float *dsdf_out, const float *grad4,
                      int32_t grad4_mode, void *stream);

/*
 * Tiny-MLP engine (RadianceNet 85-192-192-192-3, TonemapNet 33-192-3;
 * app/utils/pbr/module.py:6-39) on f32 MFMA.  `kind`: 0 radiance, 1 tonemap.
 * Weights are the reference's nn.Linear tensors ([out,in] row-major) packed by
 * esr_mlp_pack into MFMA operand order.
 */
#define ESR_MLP_RADIANCE 0   /* 85-192-192-192-3, softplus applied by the caller kernels   */
#define ESR_MLP_TONEMAP  1   /* 33-192-3                                                  */
#define ESR_MLP_BRDF     2   /* BRDFNet 76-128-128-128-5 (app/utils/pbr/module.py:42-65)  */
#define ESR_MLP_EMIT     3   /* EmissionNet 76-128-128-128-3 (app/utils/pbr/module.py:68-83) */
#define ESR_MLP_COARSE   4   /* coarse rgbnet 57-128-128-3 (app/coarse/model/voxurfc.py:134-149), 72-row input tile */
#define ESR_MLP_MAX_LAYERS 4

typedef struct esr_mlp_weights {       /* [host] struct of device pointers */
    const float *w[ESR_MLP_MAX_LAYERS];   /* [out,in] */
    const float *b[ESR_MLP_MAX_LAYERS];   /* [out]    */
} esr_mlp_weights_t;

int64_t esr_mlp_packed_floats(int kind);     /* size of the packed buffer */
int esr_mlp_pack(int kind, const esr_mlp_weights_t *w, float *packed, void *stream);
/* Every net of a step in ONE launch (n <= 8): packed32[i] <- w[i] for net kinds[i]; where packed16 != NULL and
 * packed16[i] != NULL also its bf16 twin (esr_mlp_pack_bf16).  The reference has no counterpart: its nn.Linear
 * weights are used as they are (app/utils/pbr/module.py:6-83). */
int esr_mlp_pack_batch(int n, const int32_t *kinds, const esr_mlp_weights_t *const *w, float *const *packed32,
                       void *const *packed16, void *const *packed_split, void *stream);
/*
 * Round 4: the f32 engine's MLP FORWARD on the 16-bit matrix cores with fp32 results (csrc/mlp_split.hip): every operand as
 * two fp16 planes x = x1 + x2 (x1 = fp16(x), x2 = fp16(x - x1); the weights' planes hold 64 w), a product as
 * w1.x1 + w1.x2 + w2.x1 on v_mfma_f32_32x32x16_f16 into one fp32 accumulator -- fp32-level accuracy (2e-7 .. 6e-7 of a layer's
 * largest value against double precision, as the f32 MFMA kernels) at 16/3 of the f32 matrix rate.  packed_split[i] of
 * esr_mlp_pack_batch (or NULL) receives the net's forward and transposed weights as split planes,
 * esr_mlp_packed_split_elems(kind) fp16 values.  esr_mlp_fwd_split / esr_mlp_fwd_fine_split keep the contracts of
 * esr_mlp_fwd / esr_mlp_fwd_fine (inputs X, saved tiles H and masks M, outputs z are the f32 engine's, so the f32
 * input-gradient and weight-gradient entries follow unchanged).  Kinds: ESR_MLP_RADIANCE, _TONEMAP, _BRDF, _EMIT (the coarse
 * net: ESR_EINVAL).  The reference evaluates these layers with fp32 nn.Linear (app/utils/pbr/module.py:6-83).
 */
int64_t esr_mlp_packed_split_elems(int kind);
/* fp16-element offset, inside a net's planes buffer, of its gradient gain bound G (one fp32; see esr_mlp_split_range_flag) */
int64_t esr_mlp_split_gain_offset(int kind);
/* The split kernels' range: a first plane is fp16, so an input, a hidden activation or 64 x a weight beyond 65504 would become
 * inf.  flag (device uint32, owned by the caller, sticky; NULL unregisters) is registered for the CURRENT device and ORed with 1
 * by every later (a) esr_mlp_fwd_split / esr_mlp_fwd_fine_split launch on it when an input or a hidden activation reaches
 * 60000 (or is inf; a +NaN activation) -- the output layer's results are fp32 sums that never become planes --, (b)
 * esr_mlp_pack_batch launch when fp16(64 w) of a weight is not finite (|w| >= 1023.5), or when the net's gradient gain bound
 * (below) exceeds 2^18.  The BACKWARD needs no flag: esr_mlp_pack_batch writes, behind a net's planes, the bound
 * G = max over the hidden layers of the running product of the layers' largest column sums of |W| (and 1): no hidden
 * gradient of a tile can exceed G max |dz|; esr_mlp_dgrad_split scales each tile so that G max |dz| is in [2^14, 2^15), and its
 * `amax` output (the weight-gradient kernels' scale source) is max |dz| x max(1, G / 16).  The host side
 * (esr_nerf_amd/fine_engine.py: range_probe / range_hit / f32_only) reads the flag behind the forward of every step and
 * re-runs a step that raised it on the f32 MFMA entry points, which share every buffer format.  esr_fine_plan also copies the
 * flag into bit 1 of the plan header's overflow word (informational). */
int esr_mlp_split_range_flag(uint32_t *flag);
int esr_mlp_fwd_split(int kind, const float *packed32, const void *planes, const float *X, int32_t t0, int32_t t1,
                      float *const *H, uint32_t *const *M, int save, int color_row0, float *zout, void *stream);
int esr_mlp_fwd_fine_split(const float *packed32_off, const void *planes_off, const float *packed32_emo,
                           const void *planes_emo, const float *X, int32_t t_on, int32_t t_all, float *const *H,
                           uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo, void *stream);
/* The input-gradient chain the same way (contracts of esr_mlp_dgrad / esr_mlp_dgrad_fine; the same four kinds): the split
 * buffer also holds the TRANSPOSED weights' planes.  Gradients are far below fp16's normal range, so each 32-sample tile's
 * chain runs scaled by a power of two chosen from its largest |dz| (exact), and dZ / dX are written unscaled in fp32. */
/* amax (optional, device, one float, >= 0 on entry -- normally zero): raised, with an atomic maximum, to the largest |dz| of the
 * launch's tiles x max(1, G / 16) (G: the net's gain bound, above) -- a value B with |dz| <= B and every hidden |dZ| <= 16 B;
 * esr_wgrad_job_t::amax and esr_tone_wgrad_recompute_split read it (their planes hold >= 32 B). */
int esr_mlp_dgrad_split(int kind, const void *planes, const float *dz, int32_t t0, int32_t t1, const uint32_t *const *M,
                        float *const *dZ, float *dX, float *amax, void *stream);
int esr_mlp_dgrad_fine_split(const void *planes_emo, const void *planes_off, const float *dz, int32_t t_on, int32_t t_all,
                             const uint32_t *const *M, float *const *dZ, float *dX, float *amax, void *stream);
/* out[0] = max(out[0], max |x[i]|, i < n) (device; out >= 0 on entry).  As a scale source of the split weight-gradient
 * kernels it is safe only where the caller knows its hidden gradients stay below 32 x that value (tests). */
int esr_absmax(const float *x, int64_t n, float *out, void *stream);

/*
 * Forward over tiles [t0,t1).  X: layer-1 input, tile-major [tiles,xrows,32].
 * With save != 0 every hidden layer l keeps H[l] [tiles,192,32] (read by the weight
 * gradient) and its ReLU sign bits M[l] [tiles,3,64] u32 (read by the input gradient:
 * 768 B per tile instead of 24 KB).  zout [tiles,4,32]: pre-activation outputs
 * (row 3 = 0; [tiles,8,32] for the 5-output BRDF net).  Hidden tiles are [tiles,128,32]
 * and masks [tiles,2,64] for the 128-wide nets.  color_row0 (0, 88 or 96) selects which
 * 6-row colour group of the X tile feeds the first 6 inputs.  save == 2 keeps the masks M only (for a net
 * whose weight gradient recomputes its hidden layer: esr_tone_wgrad_recompute); H may then be NULL.
 */
int esr_mlp_fwd(int kind, const float *packed, const float *X, int32_t t0, int32_t t1,
                float *const *H, uint32_t *const *M, int save, int color_row0, float *zout,
                void *stream);
/*
 * The same net over two adjacent tile ranges in ONE launch: [t0,t_mid) detached (nothing saved, colour
 * rows color_row_detached), [t_mid,t1) saved with colour rows 0.  This is the fine stage's off-net
 * (app/fine/model/voxurff.py:244-254: `.detach()` on the emissive-on rays, differentiable on the off
 * rays); RadianceNet only.
 */
int esr_mlp_fwd_mixed(int kind, const float *packed, const float *X, int32_t t0, int32_t t_mid, int32_t t1,
                      float *const *H, uint32_t *const *M, int color_row_detached, float *zout, void *stream);

/*
 * The fine stage's three radiance passes of one step in ONE launch (app/fine/model/voxurff.py:243-256): the non-emissive
 * net on tiles [0,t_on) detached (colour rows color_row_detached, nothing saved) and on [t_on,t_all) saved with colour
 * rows 0, the emissive net on [0,t_on) saved with colour rows 0.  Both nets save into the same H / M arrays (disjoint
 * tiles); z_off [t_all,4,32], z_emo [t_on,4,32].  One ramp-up and one tail instead of two launches' worth.
 */
int esr_mlp_fwd_fine(const float *packed_off, const float *packed_emo, const float *X, int32_t t_on, int32_t t_all,
                     float *const *H, uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo,
                     void *stream);
/* Its backward twin: input gradients of the emissive net on tiles [0,t_on) and of the non-emissive net on
 * [t_on,t_all) in one launch (the non-emissive net's on-tile pass is detached in the reference: no gradient). */
int esr_mlp_dgrad_fine(const float *packed_emo, const float *packed_off, const float *dz, int32_t t_on, int32_t t_all,
                       const uint32_t *const *M, float *const *dZ, float *dX, void *stream);
/* The bf16 engine's twins of the two entries above (weights of both nets as packed by esr_mlp_pack / esr_mlp_pack_bf16). */
int esr_mlp_fwd_fine_bf16(const float *packed32_off, const void *packed16_off, const float *packed32_emo,
                          const void *packed16_emo, const float *X, const void *X16, int32_t t_on, int32_t t_all,
                          float *const *H, uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo,
                          void *stream);
int esr_mlp_dgrad_fine_bf16(const void *packed16_emo, const void *packed16_off, const float *dz, int32_t t_on, int32_t t_all,
                            const uint32_t *const *M, float *const *dZ, float *dX, void *stream);

/*
 * Input/hidden gradients over tiles [t0,t1) (a NULL dZ[l] is computed but not stored).  dz [tiles,4,32] -> dZ[l] (each
 * [tiles,192,32], pre-activation grads of hidden layer l) and dX [tiles,64,32], of which the rows that lead back to a
 * grid are written, rounded up to 4: rows 0-43 for the sample nets (colour | sdf | 24 stencil taps | 12 normal
 * components = rows 0-42; the position / view-direction encodings are inputs without a gradient in the reference too),
 * rows 0-35 for the tone mapper (33 inputs), rows 0-31 for the coarse net.  The other rows are left untouched.
 */
int esr_mlp_dgrad(int kind, const float *packed, const float *dz, int32_t t0, int32_t t1,
                  const uint32_t *const *M, float *const *dZ, float *dX, void *stream);
/* The same with the grid capped at max_workgroups (0 = no cap; 256 = one workgroup per CU, which leaves half of the
 * register file and the LDS to a latency-bound kernel enqueued on another stream). */
int esr_mlp_dgrad_wg(int kind, const float *packed, const float *dz, int32_t t0, int32_t t1,
                     const uint32_t *const *M, float *const *dZ, float *dX, int32_t max_workgroups, void *stream);

/*
 * Weight/bias gradients accumulated into the reference-layout tensors gw[l] [out,in],
 * gb[l] [out] over tiles [t0,t1).  scratch: >= esr_mlp_wgrad_scratch_floats() floats of
 * device workspace for the per-workgroup partial slabs (summed by a second kernel).
 */
int64_t esr_mlp_wgrad_scratch_floats(void);
int esr_mlp_wgrad(int kind, const float *X, int color_row0, const float *const *H,
                  const float *const *dZ, const float *dz, int32_t t0, int32_t t1,
                  float *const *gw, float *const *gb, float *scratch, int64_t scratch_floats,
                  void *stream);

/*
 * Weight gradients of SEVERAL nets / tile ranges in one call.  Layers of the same kernel shape share a launch (the
 * same layer of the emissive and the non-emissive RadianceNet, both hidden layers, the 3-row output layers of all
 * nets): a weight-gradient launch has a fixed cost of ~30 us whatever its tile count, and with J jobs per launch the
 * per-workgroup partial sums that have to be exchanged shrink J-fold.  Results are those of one esr_mlp_wgrad (or
 * esr_mlp_wgrad_bf16 with bf16_operands != 0) per job.  H, dZ, gw, gb are [host] arrays of device pointers.
 */
typedef struct esr_wgrad_job {
    int32_t kind, color_row0, t0, t1;
    const float *X;
    const float *const *H;
    const float *const *dZ;
    const float *dz;
    float *const *gw;
    float *const *gb;
    /* optional (bf16 operands, ESR_MLP_RADIANCE, color_row0 == 0): the net's input tile as written by
     * esr_fine_feat_fwd_x16 -- the first-layer job then stages it like a hidden layer's tile and X is not read. */
    const void *X16;
    /* optional (f32 operands; every net kind and layer shape): device pointer to the scale source B left by
     * esr_mlp_dgrad_split / esr_mlp_dgrad_fine_split (|dz| <= B, hidden |dZ| <= 16 B; the planes hold 128 B).  Non-NULL
     * selects the split-fp16 weight-gradient kernel:
     * the same fp32 operands, every value cut into two fp16 planes on its way into the 16-bit matrix cores, fp32
     * accumulation; the gradient operand is scaled by a power of two derived from *amax (csrc/mlp.hip, SPLIT). */
    const float *amax;
} esr_wgrad_job_t;
int esr_mlp_wgrad_batch(const esr_wgrad_job_t *jobs, int32_t n_jobs, int bf16_operands, float *scratch,
                        int64_t scratch_floats, void *stream);

/*
 * Weight gradients of the tone mapper (TonemapNet 33-192-3, app/utils/pbr/module.py:24-39) from its INPUTS only:
 * Xt [tiles,48,32] (the tile esr_fine_tone_in_fwd writes) and dzt [tiles,4,32] (d loss / d pre-sigmoid output).  The
 * hidden layer is recomputed inside the kernel, so the forward need not save Ht (esr_mlp_fwd with save = 2) and the
 * input-gradient pass need not store dZt (esr_mlp_dgrad with dZ[0] = NULL): 48 KB of tile traffic per 32 samples
 * less than esr_mlp_wgrad(ESR_MLP_TONEMAP).  W0 [192,33], b0 [192], W1 [3,192]: the net's parameters in the
 * reference layout; gw0 / gb0 / gw1 / gb1 receive += the gradients.  fp32.
 */
int64_t esr_tone_wgrad_scratch_floats(void);
int esr_tone_wgrad_recompute(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                             int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1, float *gb1,
                             float *scratch, int64_t scratch_floats, void *stream);
/* The same with bf16 matrix operands (fp32 accumulation): Xt, dzt and the weights stay fp32 in memory and are rounded
 * where the bf16 engine rounds them, so neither Ht nor dZt of the tone mapper is saved by the bf16 step either. */
int esr_tone_wgrad_recompute_bf16(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                                  int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1, float *gb1,
                                  float *scratch, int64_t scratch_floats, void *stream);
/* The same with the products on the 16-bit matrix cores from split fp16 planes, fp32 results (round 4; csrc/tone_wgrad.hip):
 * amax = device pointer to esr_mlp_dgrad_split's scale source of the tone mapper's pass (|dzt| <= B, |dZt| <= 16 B). */
int esr_tone_wgrad_recompute_split(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                                   const float *amax, int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1,
                                   float *gb1, float *scratch, int64_t scratch_floats, void *stream);

/*
 * bf16 variants of the MLP engine for BASELINE.json's bf16 configurations (a build-side precision choice:
 * the reference is fp32 everywhere): bf16 MFMA OPERANDS (v_mfma_f32_32x32x16_bf16), fp32 accumulation;
 * weights are rounded by esr_mlp_pack_bf16, activations when they become an operand; biases are read from
 * the esr_mlp_pack buffer.  The tiles saved for the backward -- H[l] and dZ[l] -- are bf16, hid * 32 values per tile
 * (half the bytes of the fp32 engine's) in an engine-private order: [row / 4][32 sample slots][4 rows], sample s in
 * slot 8 ((s >> 1) & 3) + 2 (s >> 3) + (s & 1) -- written 8 bytes per lane, read back only by esr_mlp_wgrad_bf16;
 * X, zout, dz, dX, the masks M and the weight / bias gradients are fp32 with the layouts above, so the
 * feature / shading kernels are shared.  packed16: esr_mlp_packed_bf16_elems(kind) bf16 values.
 */
int64_t esr_mlp_packed_bf16_elems(int kind);
int esr_mlp_pack_bf16(int kind, const esr_mlp_weights_t *w, void *packed16, void *stream);
int esr_mlp_fwd_bf16(int kind, const float *packed32, const void *packed16, const float *X, int32_t t0,
                     int32_t t1, float *const *H, uint32_t *const *M, int save, int color_row0,
                     float *zout, void *stream);
int esr_mlp_dgrad_bf16(int kind, const void *packed16, const float *dz, int32_t t0, int32_t t1,
                       const uint32_t *const *M, float *const *dZ, float *dX, void *stream);
int esr_mlp_wgrad_bf16(int kind, const float *X, int color_row0, const float *const *H,
                       const float *const *dZ, const float *dz, int32_t t0, int32_t t1,
                       float *const *gw, float *const *gb, float *scratch, int64_t scratch_floats,
                       void *stream);

/*
 * Between the nets: lin = softplus(z_off) (+ softplus(z_emo) on on-tiles);
 * Xt [tiles,48,32] = [lin3 | sin15 | cos15 | 0...] (voxurff.py:783-788).
 */
int esr_fine_tone_in_fwd(const float *z_off, const float *z_emo, int32_t tiles_on,
                         int32_t tiles_all, float *lin, float *Xt, void *stream);

/*
 * rgb = sigmoid(zt); srgb_marched[ray] += w*rgb; lin_marched[ray] += w*lin
 * (segmented wave reduction + one atomic per ray segment; voxurff.py:258-272).
 * rgb [tiles,4,32] kept for the backward.  Outputs [n_rays,3] are accumulated
 * into (caller zero-fills).
 */
int esr_fine_composite_fwd(const float *zt, const float *lin, const int32_t *rec_ray,
                           const float *rec_w, int32_t tiles_all, float *rgb,
                           float *srgb_marched, float *lin_marched, void *stream);

/* d(srgb_marched), d(lin_marched) [n_rays,3] -> dweight [tiles*32], dzt [tiles,4,32]. */
int esr_fine_composite_bwd(const float *g_srgb, const float *g_lin, const float *rgb,
                           const float *lin, const int32_t *rec_ray, const float *rec_w,
                           int32_t tiles_all, float *dweight, float *dzt, void *stream);

/*
 * dXt [tiles,64,32] (tonemap input grads) + g_lin -> dz [tiles,4,32] (grad of the
 * radiance pre-activations: emo net on on-tiles, off net on off-tiles).  Xt [tiles,48,32]: the tile
 * esr_fine_tone_in_fwd wrote (its sin / cos rows are the factors of the encoding's derivative).
 */
int esr_fine_tone_in_bwd(const float *dXt, const float *Xt, const float *g_lin, const float *lin,
                         const float *z_off, const float *z_emo, const int32_t *rec_ray,
                         const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *dz,
                         void *stream);

/*
 * Trainer-step loss of the fine stage (app/fine/fine.py:355-382) and its
 * gradient w.r.t. the three renderer outputs, one launch.  loss [1] f32 is
 * accumulated into (caller zero-fills).
 */
int esr_fine_loss_fwd_bwd(const float *srgb_marched, const float *lin_marched,
                          const float *alphainv_last, const float *rgbs, int32_t n_rays,
                          float white_bg, float weight_linear, float weight_entropy_last,
                          float *loss, float *g_srgb, float *g_lin, float *g_last,
                          void *stream);
/* The same with every term (loss and gradients) multiplied by `scale`: a rank's share n_local / n_global of a
 * data-parallel batch, whose mean-reduced terms are means over the GLOBAL batch (no reference counterpart). */
int esr_fine_loss_fwd_bwd_dp(const float *srgb_marched, const float *lin_marched,
                             const float *alphainv_last, const float *rgbs, int32_t n_rays,
                             float white_bg, float weight_linear, float weight_entropy_last, float scale,
                             float *loss, float *g_srgb, float *g_lin, float *g_last,
                             void *stream);

/*
 * One mean-reduced two-operand term of the LTS / PDRA trainer losses and its gradients
 * (app/fine/lts.py:362-379: MSE(off, off_hat), MSE(emo, emo_hat), L1(normal, normal_eps);
 * app/fine/pdra.py:408-457: the L1 variants with separate weights for the two sides, the
 * emission suppression mean(emit_cert^2), L1(emit, emit_eps)).
 *   kind 0: mean((a-b)^2), kind 1: mean(|a-b|) over the elements of the rows with
 *   row_mask[row] == mask_value (row_mask NULL: all rows); b NULL stands for zeros.
 *   count_dev (device, optional): number of selected rows, so that a data-dependent
 *   selection (emit_cert = emit_marched[~uncert]) needs no host sync; 0 rows -> term 0.
 * loss[0] += w_value * term;  ga = w_a * d(term)/d(a);  gb = w_b * d(term)/d(b)  (NULL: skip).
 */
int esr_pair_loss_fwd_bwd(const float *a, const float *b, int64_t rows, int32_t cols,
                          const uint8_t *row_mask, int mask_value, const int32_t *count_dev,
                          int kind, float w_value, float w_a, float w_b, float *loss, float *ga,
                          float *gb, void *stream);

/* ------------------------------------------------------------------------- *
 * C. Light-transport-segment (LTS / PDRA) stage ops
 *    (ESRNeRF.forward_training, app/fine/model/esrnerf.py:486-851).
 * ------------------------------------------------------------------------- */

/*
 * Exact spatial gradient of the trilinear SDF interpolant -- replaces sample_sdf_expgrad
 * (esrnerf.py:1572-1596: autograd through differentiable_grid_sample,
 * app/utils/base/functions.py:142-309) by its closed form.  Points are either the
 * march records (rec_ray/rec_step, pts == NULL; padding records give zeros) or explicit
 * pts [n,3]; optional noise [n,3] * eps is added (the *_eps re-evaluations,
 * esrnerf.py:807-810).  out [n,4] = (sdf, d/dx, d/dy, d/dz) in WORLD xyz order.
 * zero_pad != 0: corners outside the grid contribute zero instead of being border-replicated
 * (F.grid_sample semantics of sample_sdf_grad, used for the perturbed emit/brdf evaluation).
 */
int esr_expgrad_fwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                    const int32_t *rec_ray, const int32_t *rec_step, const float *pts,
                    const float *noise, float eps, const float *sdf, int32_t n, int zero_pad,
                    float *out, void *stream);
/* g [n,4] -> atomic scatter into grad_sdf [gx,gy,gz] (the op is linear in the grid). */
int esr_expgrad_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                    const int32_t *rec_ray, const int32_t *rec_step, const float *pts,
                    const float *noise, float eps, const float *g, int32_t n, int zero_pad,
                    float *grad_sdf, void *stream);

/*
 * The light-transport step's glue (esrnerf.py:781-830: reference-order bookkeeping, surface-point gathers, perturbed
 * positions), fused: four launches for what are ~60 small torch ops in the reference.
 *   esr_lts_ref_order: perm[pos] = j for every compact slot j with rec_ray[j] >= 0, pos = its rank in the reference's
 *       ray-sorted sample order (cnt3_cumsum = inclusive int64 cumsum of cnt3); ray64[j] = (int64) rec_ray[j].
 *   esr_lts_perturb:   pts_e[k] = pts_all[perm[k]] + noise_emit[k] * emit_eps,  noise_n[perm[k]] = noise_normal[k]
 *       (noise_n [n_slots,3] zeroed by the caller).
 *   esr_lts_gather_rows: out[k][c] = src(row perm[k] -- or k when perm is NULL --, column col0 + c); src tile-major
 *       [tiles][tile_rows][32] when tile_rows > 0, else row-major with row_stride floats per row.
 *   esr_lts_gather_points: per surface point (compact slot jp[p]) position / view direction / SDF (each written twice:
 *       pts2, sdf2 [2P], vd2 rows < P), unit normal, material heads, uncertainty mask.
 */
int esr_lts_ref_order(const int32_t *rec_ray, const int32_t *cnt3, const int32_t *off3, const int64_t *cnt3_cumsum,
                      int32_t n_slots, int64_t *perm, int64_t *ray64, void *stream);
int esr_lts_perturb(const float *pts_all, const int64_t *perm, const float *noise_normal, const float *noise_emit,
                    float emit_eps, int32_t m3, float *noise_n, float *pts_e, void *stream);
int esr_lts_gather_rows(const float *src, int32_t tile_rows, int32_t row_stride, int32_t col0, int32_t n_ch,
                        const int64_t *perm, int32_t n, float *out, void *stream);
typedef struct esr_lts_gather {
    const int64_t *jp, *ray64;
    const float *pts_all, *eg, *rec_sdf, *viewdirs, *brdf_a, *emit_a;
    const uint8_t *umask_rays;
    int32_t n_pts;
    float *pts2, *vd2, *sdf2, *normal, *base, *rough, *metal, *emis;
    uint8_t *umask;
    int32_t *pt1;      /* optional: pt1[jp[p]] = p + 1 (int32 [n_slots], zeroed by the caller): slot -> surface point */
} esr_lts_gather_t;
int esr_lts_gather_points(const esr_lts_gather_t *g, void *stream);

/*
 * Hemisphere directions -- replaces diffuse_scattering (app/utils/pbr/functions.py:10-18)
 * given the standard-normal draws: normalise, flip into the hemisphere of `normal`.
 * raw, dirs [n_pts, rays_plus_one, 3]; normal [n_pts,3].
 */
int esr_lts_dirs(const float *raw, const float *normal, int32_t n_pts, int32_t rays_plus_one,
                 float *dirs, void *stream);

/*
 * Rendering-equation estimate at n_pts surface points with n_rays secondary rays each
 * (esrnerf.py:565-572 disney_reflection, :653-677 env term, means, emo_hat; pbr/module.py:133-143
 * spherical-Gaussian env map; pbr/functions.py:108-173).  The two "copies" are the camera view
 * direction and the random one (-dirs[:, n_rays]).  All arrays row-major device memory.
 */
typedef struct esr_lts_args {
    int32_t n_pts, n_rays, n_sg, pdra_mode;
    const float *base, *rough, *metal;      /* [P,3] [P] [P]  BRDF parameters at the points      */
    const float *normal, *view;             /* [P,3] unit normal (detached), [P,3] view direction */
    const float *dirs;                      /* [P, n_rays+1, 3] from esr_lts_dirs                 */
    const float *off_m, *emo_m, *last2;     /* [P*R,3] [P*R,3] [P*R] marched secondary rays       */
    const float *mus, *lambdas, *lobes;     /* [J,3] [J] [J,3] envmap parameters (J <= 64)        */
    const float *emission;                  /* [P,3]                                              */
    const uint8_t *umask;                   /* [P] uncertain-ray flags (pdra) or NULL             */
} esr_lts_args_t;

typedef struct esr_lts_grads {
    float *d_off_m, *d_emo_m, *d_last2;     /* [P*R,3] [P*R,3] [P*R]  written                     */
    float *d_base, *d_rough, *d_metal, *d_emission;   /* [P,3] [P] [P] [P,3]  written             */
    float *d_mus, *d_lambdas, *d_lobes;     /* accumulated into (caller zero-fills)               */
} esr_lts_grads_t;

/* off_hat, emo_hat [2P,3]: rows [0,P) camera direction, [P,2P) random direction. */
int esr_lts_combine_fwd(const esr_lts_args_t *args, float *off_hat, float *emo_hat, void *stream);
int esr_lts_combine_bwd(const esr_lts_args_t *args, const float *g_off_hat, const float *g_emo_hat,
                        const esr_lts_grads_t *grads, void *stream);

/*
 * Emission edit of the re-lighting fine-tune, in place on emit [n,3] -- replaces
 * app/fine/model/esrnerf.py:427-441 with rgb_to_hsv / hsv_to_rgb of
 * app/utils/pbr/functions.py:214-255.  em_modes int64 [n]: 0 emission off, 1 unchanged,
 * 2 scaled by em_intensities [n], 3 hue/saturation replaced by em_colors [n,2] (value kept),
 * 4 both.
 */
int esr_emit_edit(float *emit, const int64_t *em_modes, const float *em_intensities,
                  const float *em_colors, int32_t n, void *stream);

/*
 * Small tile-major helpers of the LTS renderer.
 *  esr_act_*: out = act(z) on the first n_ch rows of [tiles,rows,32] tiles, rest zero
 *     (act 0 softplus: radiance / emission heads, 1 sigmoid: BRDF head; pbr/module.py:21,64,83).
 *  esr_composite3_*: out[ray,c] += w * v[c] (segment sum over sorted rays, esrnerf.py:639-651,783-788)
 *     and its backward (accumulate bit 0: add into dv, bit 1: add into dweight).
 *  esr_lts_tone_in_bwd: as esr_fine_tone_in_bwd for lin = off + emo WITHOUT the detach
 *     (esrnerf.py:751-757): dz_off on all tiles, dz_emo on the on-tiles.
 *  esr_sample_points: world positions of the march records (padding -> 0).
 */
/*
 * Round 4: the same glue with FEWER launches (a light-transport step issued ~100 kernels shorter than 10 us; what they
 * cost is the host's enqueue time, tools/host_profile.py).  Batched forms: one launch, blockIdx.y = job.
 *
 * esr_act_batch: up to ESR_ACT_MAX_JOBS activation jobs.  A forward job is esr_act_fwd.  A backward job computes
 *   dz[slot][c] = act'(z) * ( g_tile[slot][c]                                  (tile-major, optional)
 *                           + src[k][c]       k = inv ? inv[slot] : slot, if 0 <= k < n_src and c < src_c   (optional)
 *                           + ex_i[p][c - ex_col0[i]]   p = pt1[slot] - 1 >= 0, for up to three extras       (optional) )
 *   i.e. the reference-order gradient rows of a head (`src`, scattered back through the inverse of esr_lts_ref_order's
 *   permutation), the rendering equation's gradients at the chosen surface points (`ex`, through pt1 of
 *   esr_lts_gather_points) and a tile-major upstream gradient, summed and multiplied by the activation's derivative in
 *   one pass -- the reference's lines are index_put / index_add_ / cat on [M3, C] tensors followed by the activation's
 *   autograd (app/fine/model/esrnerf.py:770-806, backward of :792-851).
 */
#define ESR_ACT_MAX_JOBS 4
typedef struct esr_act_job {
    const float *z;            /* [tiles, rows, 32] pre-activations                                        */
    const float *g_tile;       /* backward: tile-major upstream gradient or NULL                           */
    float *out;                /* [tiles, rows, 32]                                                        */
    int32_t tiles, rows, n_ch, act, bwd;
    const float *src;          /* backward: [n_src, src_c] row-major or NULL                               */
    int32_t src_c, n_src;
    const int32_t *inv;        /* slot -> row of src (-1: none); NULL: identity                            */
    const int32_t *pt1;        /* slot -> surface point + 1 (0: none) or NULL                              */
    const float *ex[3];        /* [P, ex_c[i]] row-major or NULL                                           */
    int32_t ex_c[3], ex_col0[3];
} esr_act_job_t;
int esr_act_batch(const esr_act_job_t *jobs, int32_t n_jobs, void *stream);

/* esr_lts_gather_rows for up to ESR_GATHER_MAX_JOBS (src, out) pairs in one launch. */
#define ESR_GATHER_MAX_JOBS 4
typedef struct esr_gather_job {
    const float *src;
    int32_t tile_rows, row_stride, col0, n_ch;
    const int64_t *perm;
    int32_t n;
    float *out;
} esr_gather_job_t;
int esr_lts_gather_rows_batch(const esr_gather_job_t *jobs, int32_t n_jobs, void *stream);

/* esr_pair_loss_fwd_bwd for up to ESR_PAIR_MAX_JOBS terms in one launch (all add into the same `loss`). */
#define ESR_PAIR_MAX_JOBS 6
typedef struct esr_pair_job {
    const float *a, *b;
    int64_t rows;
    int32_t cols;
    const uint8_t *row_mask;
    int32_t mask_value;
    const int32_t *count_dev;
    int32_t kind;
    float w_value, w_a, w_b;
    float *ga, *gb;
} esr_pair_job_t;
int esr_pair_loss_batch(const esr_pair_job_t *jobs, int32_t n_jobs, float *loss, void *stream);

/* esr_lts_ref_order that also writes the INVERSE: inv[j] = rank of compact slot j in the reference's order, -1 for a
 * padding slot (inv: int32 [n_slots]). */
int esr_lts_ref_order_inv(const int32_t *rec_ray, const int32_t *cnt3, const int32_t *off3, const int64_t *cnt3_cumsum,
                          int32_t n_slots, int64_t *perm, int64_t *ray64, int32_t *inv, void *stream);

/* esr_lts_dirs that also writes what the secondary march and the points pass read (esrnerf.py:565-591): the secondary
 * rays' origins o2 [P * n_rays, 3] (pts[p] repeated), their directions d2 [P * n_rays, 3] (dirs[:, :n_rays]) and the
 * random view direction v_rand [P, 3] = -dirs[:, n_rays]; pts [P, 3].  rays_plus_one = n_rays + 1. */
int esr_lts_dirs_rays(const float *raw, const float *normal, const float *pts, int32_t n_pts, int32_t rays_plus_one,
                      float *dirs, float *o2, float *d2, float *v_rand, void *stream);

int esr_act_fwd(const float *z, int32_t tiles, int32_t rows, int32_t n_ch, int act, float *out, void *stream);
int esr_act_bwd(const float *z, const float *g, int32_t tiles, int32_t rows, int32_t n_ch, int act,
                float *dz, void *stream);
int esr_composite3_fwd(const float *v, int32_t rows, const int32_t *rec_ray, const float *rec_w,
                       int32_t tiles, float *out, void *stream);
int esr_composite3_bwd(const float *g, const float *v, int32_t rows, const int32_t *rec_ray,
                       const float *rec_w, int32_t tiles, int accumulate, float *dv, float *dweight,
                       void *stream);
int esr_lts_tone_in_bwd(const float *dXt, const float *Xt, const float *g_lin, const float *lin, const float *z_off,
                        const float *z_emo, const int32_t *rec_ray, const float *rec_w, int32_t tiles_on,
                        int32_t tiles_all, float *dz_off, float *dz_emo, void *stream);
int esr_sample_points(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                      const int32_t *rec_ray, const int32_t *rec_step, int32_t n, float *pts, void *stream);

/* ------------------------------------------------------------------------- *
 * D. Coarse stage (VoxurfC) -- dense whole-grid operators
 * ------------------------------------------------------------------------- */

/*
 * Gaussian smoothing of the SDF grid -- replaces Gaussian3DConv.forward
 * (app/utils/base/module.py:145-177: nn.Conv3d(1,1,k, padding=k//2, padding_mode="replicate"),
 * called on every forward at app/coarse/model/voxurfc.py:202).  in/out [gx,gy,gz];
 * weights_host: k*k*k floats in HOST memory in Conv3d order (they travel as a kernel argument).
 * _bwd is the exact adjoint (a gather, no atomics): gin += conv^T(gout).
 */
int esr_gauss3d_fwd(const float *in, const float *weights_host, int ksize, int32_t gx, int32_t gy,
                    int32_t gz, float *out, void *stream);
int esr_gauss3d_bwd(const float *gout, const float *weights_host, int ksize, int32_t gx, int32_t gy,
                    int32_t gz, float *gin, void *stream);

/*
 * Dense central-difference gradient of the SDF grid -- replaces VoxurfC.neus_sdf_gradient
 * (app/coarse/model/voxurfc.py:597-616).  grad is channels-last [gx,gy,gz,3] (component a =
 * d/d grid axis a = world x,y,z), zero on the boundary layer of each axis.  _bwd: gsdf += adjoint.
 */
int esr_central_grad_fwd(const float *sdf, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                         float *grad, void *stream);
int esr_central_grad_bwd(const float *ggrad, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                         float *gsdf, void *stream);

/*
 * Fused march of the coarse renderer (app/coarse/model/voxurfc.py:186-219): sampler -> in-box ->
 * mask cache -> SDF tap of the SMOOTHED grid -> NeuS "interp" alpha -> alpha2weight over ALL
 * mask-cache survivors -> weight > thres -> alpha2weight AGAIN over the survivors (weights,
 * alphainv_last and cum_weights = sum of weights come from this second pass).  Same three-call
 * protocol and record layout as esr_fine_march_*; plan.m2 == plan.m1 (there is no alpha mask).
 */
int esr_coarse_march_count(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                           const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                           int32_t *cnt3, float *alphainv_last, float *cum_weights, int32_t *ray_stats,
                           esr_plan_t *plan, void *stream);
int esr_coarse_march_fill(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                          const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                          const int32_t *off3, int32_t *rec_ray, int32_t *rec_step, float *rec_w,
                          float *rec_sdf, void *stream);
int esr_coarse_march_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                         const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                         const int32_t *off3, const float *dweight, const float *dlast,
                         float *grad_sdf_smooth, void *stream);

/*
 * The coarse march under cfg neus_alpha: "grad" (app/coarse/model/voxurfc.py:171-174, 204-210;
 * app/utils/base/functions.py:45-69): section SDFs extrapolated with the trilinear sample of the dense gradient grid
 * gg [X,Y,Z,3] (esr_central_grad_fwd of the unsmoothed SDF grid) along the batch's view directions.  The backward adds
 * d/d gg into grad_gg [X,Y,Z,3] (zero-initialised or already holding the normal features' share), which
 * esr_central_grad_bwd folds into the SDF gradient.
 */
int esr_coarse_march_count_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                              const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                              int32_t *cnt3, float *alphainv_last, float *cum_weights, int32_t *ray_stats,
                              esr_plan_t *plan, void *stream);
int esr_coarse_march_fill_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                             const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                             const int32_t *off3, int32_t *rec_ray, int32_t *rec_step, float *rec_w, float *rec_sdf,
                             void *stream);
int esr_coarse_march_bwd_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                            const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                            const int32_t *off3, const float *dweight, const float *dlast, float *grad_sdf_smooth,
                            float *grad_gg, void *stream);

/*
 * Per-sample features of the coarse renderer (app/coarse/model/voxurfc.py:221-250) on the march
 * records.  grad_grid [gx,gy,gz,3] (esr_central_grad_fwd), off_color / emo_color [gx,gy,gz,12]
 * channels-last.  X [tiles,72,32] rows: 0-11 off colour | 12-23 emo colour (on-tiles) | 24-26 normal
 * = g/(|g|+1e-5) | 27-29 xyz | 30-44 sin | 45-59 cos | 60-68 view PE | 69-71 zero; gnorm [tiles*32] = |g|.
 * _bwd: dX_off (all tiles) / dX_emo (on-tiles) [tiles,64,32] from esr_mlp_dgrad(ESR_MLP_COARSE) ->
 * atomic scatters into the colour-grid gradients and into g_grad_grid [gx,gy,gz,3].
 */
int esr_coarse_feat_fwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                        const float *viewdirs, const int32_t *rec_ray, const int32_t *rec_step,
                        int32_t tiles_on, int32_t tiles_all, const float *grad_grid,
                        const float *off_color, const float *emo_color, float *X, float *gnorm,
                        void *stream);
int esr_coarse_feat_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                        const float *viewdirs, const int32_t *rec_ray, const int32_t *rec_step,
                        int32_t tiles_on, int32_t tiles_all, const float *X, const float *gnorm,
                        const float *dX_off, const float *dX_emo, float *g_grad_grid,
                        float *g_off_color, float *g_emo_color, void *stream);

/*
 * rgb = sigmoid(z_off) + [on-tiles] sigmoid(z_emo) and its weighted segment sum per ray
 * (voxurfc.py:240-258; srgb_marched [n_rays,3] accumulated into, caller zero-fills); backward
 * with white_bg = 1 - sum of weights folded in: dweight = g_srgb[ray].rgb - g_white_bg[ray].
 */
int esr_coarse_shade_fwd(const float *z_off, const float *z_emo, const int32_t *rec_ray,
                         const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *rgb,
                         float *srgb_marched, void *stream);
int esr_coarse_shade_bwd(const float *g_srgb, const float *g_white_bg, const float *rgb,
                         const float *z_off, const float *z_emo, const int32_t *rec_ray,
                         const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *dz_off,
                         float *dz_emo, float *dweight, void *stream);

/* ------------------------------------------------------------------------- *
 * F. Image rendering (forward_evaluate) helpers
 * ------------------------------------------------------------------------- */

/*
 * Per-sample auxiliaries of forward_evaluate (app/fine/model/voxurff.py:431-443, app/coarse/model/voxurfc.py:
 * 395-412) from a feature tile X [tiles,xrows,32] whose rows row_nx/ny/nz hold the unit normal (fine tile:
 * 40, 36, 32 = the radius-1 finite difference; coarse tile: 24, 25, 26): aux [tiles,8,32] rows 0-2 =
 * camera-space normal colour ((n @ pos_rt) * (1,-1,-1) + 1) / 2, row 4 = step_id * stepdist, other rows 0.
 * pos_rt_host: 3x3 row-major in HOST memory.  Composite with esr_composite3_fwd(aux, 8, ...) and
 * esr_composite3_fwd(aux + 4*32, 8, ...) (column 0 of the second result is the depth).
 * esr_eval_disp: depth[i] = depth3[3i]; disp[i] = 1 / (depth + alphainv_last * far).
 */
int esr_eval_aux(const float *X, int32_t xrows, int32_t row_nx, int32_t row_ny, int32_t row_nz,
                 const int32_t *rec_ray, const int32_t *rec_step, int32_t tiles,
                 const float *pos_rt_host, float stepdist, float *aux, void *stream);
int esr_eval_disp(const float *depth3, const float *alphainv_last, float far_, int32_t n_rays,
                  float *depth, float *disp, void *stream);

/* ------------------------------------------------------------------------- *
 * E. Optimizer
 * ------------------------------------------------------------------------- */

/*
 * One fused Adam update of a parameter tensor, in place -- replaces `adam`
 * (app/utils/optimizer.py:183-228; betas (0.9, 0.99) at :60, optional per-voxel lr :98-100).
 * step >= 1 is the value AFTER the increment (bias corrections 1 - beta^step).  All pointers:
 * n contiguous fp32 device values; per_lr NULL or [n].
 */
int esr_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                  const float *per_lr, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, void *stream);

/*
 * Smoothed-gradient TV term of the fine / lts trainers, forward and backward -- replaces the dense torch chain
 * neus_sdf_gradient (app/fine/model/voxurff.py:723-742) -> GradientConv (app/utils/base/module.py:180-211, 3x3x3,
 * replicate padding; the conv branch is detached) -> masked mean of squares (voxurff.py:609-617).
 *   fwd: loss[0] += weight * mean_{c, p in mask} (conv(g_c)[p] - g_c[p])^2 ;  work6 [6, X, Y, Z] keeps g and err
 *   bwd: grad_sdf += grad_out * d loss / d sdf  (from the err planes left in work6 by fwd)
 * sdf [X,Y,Z]; mask [X,Y,Z] bytes (nonempty_mask); conv_w27: HOST array, kernel order [dx][dy][dz];
 * masked_cells = number of set mask bytes (the mean runs over 3 * masked_cells values).
 */
int esr_smooth_grad_tv_fwd(const float *sdf, const uint8_t *mask, const float *conv_w27, float conv_bias,
                           int32_t gx, int32_t gy, int32_t gz, float voxel_size, int64_t masked_cells,
                           float weight, float *work6, float *loss, void *stream);
int esr_smooth_grad_tv_bwd(const float *work6, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                           int64_t masked_cells, float weight, const float *grad_out /* device scalar or NULL = 1 */,
                           float *grad_sdf, void *stream);

/*
 * HOST helper (no device work, no stream): out[0..k) = np.random.choice(n, k, replace=False) of numpy's legacy
 * RandomState (the surface-point draw of app/fine/model/esrnerf.py:792 / :470), bit for bit, on the MT19937 state
 * passed in (key: 624 words, *pos: 0..624; both updated).  Exists so that the draw can run on a worker thread
 * without the GIL while the primary pass is being enqueued.
 */
int esr_host_choice_noreplace(uint32_t *key, int32_t *pos, int64_t n, int64_t k, int64_t *out);
/*
 * The same draw on a worker thread owned by the library (one thread, jobs in submission order): _start returns at once with a
 * job handle, _wait blocks until that job is done, returns its code and releases the handle (exactly one _wait per _start;
 * key / pos / out must stay valid until then).  No interpreter involvement on the worker: starting it costs microseconds.
 */
int esr_host_choice_start(uint32_t *key, int32_t *pos, int64_t n, int64_t k, int64_t *out, void **job);
int esr_host_choice_wait(void *job);

/* ------------------------------------------------------------------------- *
 * F. Data-parallel gradient exchange (no reference counterpart: the reference is single-process,
 *    SURVEY 2a / 8(e); the sum over ranks itself is torch.distributed = RCCL)
 * ------------------------------------------------------------------------- */

/*
 * Brick-sparse view of a flat fp32 gradient buffer (16-byte aligned, n values): bricks of
 * esr_brick_floats() (= 128) consecutive values; the last brick may be ragged.
 *   esr_brick_flags:  flags[b] = 1 if any value of brick b is non-zero, else 0  (ceil(n/128) bytes)
 *   esr_brick_pack:   packed[k*128 .. ] = brick brick_idx[k] (ragged tail zero-filled), k < n_idx
 *   esr_brick_unpack: the inverse scatter (values past n are dropped)
 * brick_idx: int64 device array, strictly increasing brick numbers < ceil(n/128).
 */
int esr_brick_floats(void);
int esr_brick_flags(const float *buf, int64_t n, uint8_t *flags, void *stream);
int esr_brick_pack(const float *buf, int64_t n, const int64_t *brick_idx, int64_t n_idx, float *packed,
                   void *stream);
int esr_brick_unpack(const float *packed, const int64_t *brick_idx, int64_t n_idx, float *buf, int64_t n,
                     void *stream);
/*
 * The union's brick list on the device: idx[0..cap) = the first `cap` flagged bricks in ascending order, unused slots -1
 * (what esr_brick_pack / _unpack skip); *count = the number of flagged bricks, which may exceed cap (the caller reads it
 * later and sends the overflow in a second pass).  scratch: esr_brick_list_scratch_ints() int32.  Replaces six torch
 * launches inside the data-parallel step (esr_nerf_amd/grad_sync.py).
 */
int64_t esr_brick_list_scratch_ints(void);
int esr_brick_list(const uint8_t *flags, int64_t n_bricks, int64_t cap, int64_t *idx, int64_t *count, int32_t *scratch,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ESR_HIP_H */
