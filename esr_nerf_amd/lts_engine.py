"""Host driver of the LTS / PDRA renderer (``ESRNeRF.forward_training``, reference:
app/fine/model/esrnerf.py:486-851) on the kernels of libesr_hip.so.

Three sampling passes share one set of kernels:
  primary   the camera rays (march records)        -> srgb/lin/emit composites, per-sample heads
  points    the P chosen surface samples, twice    -> "lin/pbr/off", "lin/pbr/emo" (explicit points)
  secondary P x R hemisphere rays (march records)  -> incoming radiance for the rendering equation
and ``esr_lts_combine_*`` ties them together (env map, Disney reflection, hemisphere means).

Arithmetic is in HIP kernels; torch is used for memory, for re-ordering per-sample tensors
between the kernels' compact tile order and the reference's ray-sorted order (index_select /
index_add on small arrays), for the random draws (``torch.randn`` / ``np.random.choice`` exactly
where the reference draws them) and for the autograd edge.
"""
from __future__ import annotations


import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .fine_engine import DX_ROWS, KIND_RADIANCE, KIND_TONEMAP, X_ROWS, XT_ROWS, FineEngine

KIND_BRDF, KIND_EMIT = 2, 3
ACT_SOFTPLUS, ACT_SIGMOID = 0, 1


class Pass:
    """Records + tile-major workspace of one sampling pass (grow-only)."""

    def __init__(self, device, name):
        self.device, self.name = device, name
        self.cap = 0
        self.bufs: Dict[str, torch.Tensor] = {}
        self.rows: Dict[str, int] = {}
        self.tiles_on = self.tiles_all = 0
        self.n_rays = 0
        self.counts: Dict[str, int] = {}
        self.fa = None            # esr_feat_args_t
        self.keep: List[torch.Tensor] = []   # tensors referenced by raw pointers

    def ensure(self, tiles):
        # headroom from the FIRST allocation on: the tile counts of the light-transport passes move by +-15 % from step to step
        # (random surface points and directions), and a step that exceeds the capacity reallocates every buffer of the pass --
        # 12 hipMalloc calls, 82 ms, once (tools/debug/alloc_per_step.py: step 6 of a C5 run, in the middle of a short bench run)
        if tiles > self.cap:
            self.cap = int(max(tiles, self.cap) * 1.25) + 16
            self.bufs = {}

    def buf(self, name, rows=1, dtype=torch.float32):
        t = self.bufs.get(name)
        if t is None:
            t = torch.empty(self.cap * rows * 32, dtype=dtype, device=self.device)
            self.bufs[name] = t
            self.rows[name] = rows
        return t

    def rowmajor(self, name, tiles=None):
        """[tiles*32, rows] copy of a tile-major buffer."""
        tiles = self.tiles_all if tiles is None else tiles
        r = self.rows[name]
        return self.bufs[name][: tiles * r * 32].view(tiles, r, 32).permute(0, 2, 1).reshape(tiles * 32, r)

    def from_rowmajor(self, name, rows, x):
        """write a [tiles*32, c<=rows] row-major tensor into a tile-major buffer (rest zero)."""
        tiles = x.shape[0] // 32
        t = self.buf(name, rows)
        v = t[: tiles * rows * 32].view(tiles, rows, 32)
        v.zero_()
        v[:, : x.shape[1], :] = x.view(tiles, 32, x.shape[1]).permute(0, 2, 1)
        return t


@dataclass
class LtsCtx:
    scene: object
    scene2: object
    batch: Dict[str, torch.Tensor]
    perm: torch.Tensor                 # compact index of every surviving sample, in reference order
    jp: torch.Tensor                   # compact indices of the LTS points
    n_pts: int
    n_2nd: int
    pdra: bool
    t: Dict[str, torch.Tensor] = field(default_factory=dict)
    eps: Dict[str, float] = field(default_factory=dict)
    f32_only: bool = False             # the forward ran on the f32 MFMA kernels (range fallback): so must the backward


class _PointDraw:
    """``np.random.choice(n, k, replace=False)`` of numpy's GLOBAL legacy generator (the surface-point draw of
    esrnerf.py:792), started on a worker thread as soon as ``n`` -- the survivor count -- is known and collected where
    the reference draws it.  The draw is a full shuffle of range(n) on the host (1.3-1.6 ms at C4, 12 ms at 600 k
    survivors) and the light-transport pass cannot be enqueued without it; beside the enqueueing of the primary pass
    it is free.  ``esr_host_choice_noreplace`` is numpy's algorithm on numpy's state, bit for bit, called through
    a worker thread of the library (no GIL).  It advances numpy's global state IN PLACE between the constructor and
    ``result()``: nothing else may draw from ``np.random`` in between (nothing on this path does)."""

    def __init__(self, n: int, k: int, ring: list):
        """``ring``: the calling engine's [buffer, buffer, last slot, capacity] -- two alternating pinned buffers sized
        for the engine's LARGEST draw and sliced to k (one ring per engine: a class-level dict keyed by k grew by two
        pinned buffers for every distinct survivor-limited k and was shared by every engine of the process).
        The worker is a thread of the library (esr_host_choice_start / _wait): handing the call to a concurrent.futures
        worker cost this thread ~0.1 ms (submit + the worker taking the interpreter lock) in the one segment of the step
        where the device waits for the host."""
        # numpy's state is used IN PLACE (np.random.get_state / set_state cost this thread 33 + 35 us per step)
        key_addr, pos_addr = _np_global_mt19937()
        # pinned: the upload in lts_forward must not block the host (a pageable copy waits for the stream to drain,
        # after which every small launch of the light-transport glue shows its full launch latency: ~0.5 ms idle per step)
        # (two alternating buffers, allocated once: a pinned allocation per step cost the host ~60 us right after
        # the plan read, with the device idle -- tools/trace_lts.sh)
        if k > ring[3]:
            ring[0] = ring[1] = None
            ring[3] = k
        slot = ring[2] = ring[2] ^ 1
        if ring[slot] is None:
            ring[slot] = torch.empty(ring[3], dtype=torch.int64, pin_memory=torch.cuda.is_available())
        self._out_t = ring[slot][:k]
        self._job = C.c_void_p(0)
        # numpy's own lock on the generator for the whole start-to-wait window: the worker mutates the MT19937 state in place,
        # and any other Python thread that draws from np.random meanwhile (a data loader) would race on raw memory.
        # RULE for callers (lts_forward / finetune_forward hold this object from the point draw to its result()): the lock is
        # NOT re-entrant -- no np.random.* call on the calling thread inside that window (it would deadlock; a `prelude`
        # callback runs before the window opens); other threads' np.random calls stall for the ~0.3 ms of the draw.
        bg = getattr(getattr(np.random.mtrand, "_rand", None), "_bit_generator", None)
        if bg is None or not hasattr(bg, "lock"):
            raise RuntimeError("numpy's global MT19937 exposes no lock (np.random.mtrand._rand._bit_generator.lock): this numpy "
                               "version is not supported by the in-place point draw")
        self._lock = bg.lock
        self._lock.acquire()
        try:
            _lib.check(_lib.lib().esr_host_choice_start(C.c_void_p(key_addr), C.c_void_p(pos_addr), C.c_int64(n), C.c_int64(k),
                                                        C.c_void_p(self._out_t.data_ptr()), C.byref(self._job)), "esr_host_choice_start")
        except Exception:
            self._job = None
            self._lock.release()
            raise

    def result(self) -> torch.Tensor:
        self.close()
        return self._out_t

    def close(self):
        """Wait for the worker and release numpy's lock (idempotent; also from __del__: an exception between the
        constructor and result() -- a march overflow in the secondary pass -- must not leave the worker writing numpy's
        state and the pinned ring buffer behind a dropped object)."""
        job, self._job = self._job, None
        if job is None:
            return
        try:
            _lib.check(_lib.lib().esr_host_choice_wait(job), "esr_host_choice_noreplace")
        finally:
            self._lock.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_NP_MT = None


def _np_global_mt19937():
    """(address of key[624], address of pos) inside numpy's GLOBAL legacy generator (``np.random.seed`` / ``set_state`` rewrite
    that state in place, so the addresses hold for the life of the process).  numpy publishes the address of its
    ``mt19937_state {uint32_t key[624]; int pos;}`` through the bit generator's ctypes interface; the layout is verified
    against ``get_state()`` the first time (and once more after a perturbation of the state)."""
    global _NP_MT
    if _NP_MT is None:
        bg = np.random.mtrand._rand._bit_generator
        if type(bg).__name__ != "MT19937":
            raise RuntimeError("numpy's global generator is not the legacy MT19937")
        addr = bg.ctypes.state_address
        addr = int(getattr(addr, "value", addr))

        def same():
            st = np.random.get_state()
            key = np.frombuffer((C.c_uint32 * 624).from_address(addr), dtype=np.uint32)
            return st[0] == "MT19937" and np.array_equal(key, st[1]) and C.c_int.from_address(addr + 624 * 4).value == int(st[2])
        ok = same()
        st0 = np.random.get_state()
        np.random.random(3)                    # (moves pos, or refills the block)
        ok = ok and same()
        np.random.set_state(st0)
        if not (ok and same()):
            raise RuntimeError("numpy's MT19937 state is not laid out as {uint32 key[624]; int pos;} at its published address")
        _NP_MT = (bg, addr, addr + 624 * 4)
    if np.random.mtrand._rand._bit_generator is not _NP_MT[0]:
        raise RuntimeError("numpy's global generator object was replaced")
    return _NP_MT[1], _NP_MT[2]


def fibonacci_hemisphere(count: int) -> torch.Tensor:
    """``count`` unit vectors on the upper hemisphere (z >= 0) along a golden-angle spiral: the upper half of a
    2*count-point Fibonacci sphere, k = count .. 2*count-1, azimuth = golden_angle * (k + 1), cos(polar) =
    (k + 0.5) / count - 1 (pbr/functions.py:176-194 with random = False, up = True).  fp32 arithmetic on the host, as in the reference (an
    int64 arange times python floats promotes to torch's default float32): the table is bit-identical to its."""
    k = torch.arange(count, 2 * count)
    ga = math.pi * (3.0 - math.sqrt(5.0))
    phi = ga * ((k + 1.0) % (2 * count))
    cos_t = ((k + 0.5) * (1.0 / count)) - 1.0
    sin_t = torch.sqrt(1.0 - cos_t * cos_t)
    return torch.stack([torch.cos(phi) * sin_t, torch.sin(phi) * sin_t, cos_t], dim=-1)


class LtsEngine(FineEngine):
    def __init__(self, device, mlp_dtype: str = "f32"):
        super().__init__(device, mlp_dtype)
        self.ray_sampling = "random"        # or "fib" (cfg.app.model.ray_sampling; esrnerf.py:188-192)
        self._draw_ring = [None, None, 0, 0]   # pinned buffers of the surface-point draw (_PointDraw)
        self.zero_arena = True
        self.prim = Pass(self.device, "primary")
        self.pts = Pass(self.device, "points")
        self.sec = Pass(self.device, "secondary")
        self.epsp = Pass(self.device, "eps")
        self._wgrad_jobs = None
        # flush points of the batched weight gradients inside the backward (_flush_wgrad; empty = all at the end).  C5 pdra
        # bf16, 100 steps x 3 on one box: none 3.45 ms, {1,2} 3.39, {1,2,3} 3.33; with 4 and 5 too: -0.6 %, inside the noise,
        # for two more batched calls per step -- not taken.
        self.wgrad_early = {1, 2, 3}
        # the passes whose grid scatters leave the main stream (_on_scatter_stream; empty = none).  C5 pdra bf16, 100 steps x 3
        # on one box: none 3.29 ms, {1} 3.18; C4 lts f32: 3.99 -> 3.84.  {1,2}: inside the noise of {1}.
        self.eps_stream = True              # (lts_forward: the perturbed heads' pass on a stream of its own)
        self.scatter_streamed = {1}
        self.last_draws = None              # the random draws of the last lts_forward (the range fallback replays them)
        for k, kind in (("brdf", KIND_BRDF), ("emit", KIND_EMIT)):
            self.packed[k] = torch.empty(self.L.esr_mlp_packed_floats(kind), dtype=torch.float32,
                                         device=self.device)

    def _scatter_draws(self, n_pts: int, count: int) -> torch.Tensor:
        """Un-normalised scattering directions [n_pts, count, 3] that ``esr_lts_dirs`` normalises and flips into each
        point's hemisphere: standard-normal draws (``diffuse_scattering``, pbr/functions.py:10-18) or, with
        ``ray_sampling: fib``, the same Fibonacci spiral for every point (``diffuse_scattering_fib``, :21-32)."""
        if self.ray_sampling == "fib":
            return fibonacci_hemisphere(count).to(self.device).float().expand(n_pts, count, 3).contiguous()
        return torch.randn(n_pts, count, 3, device=self.device)

    # ------------------------------------------------------------------ zero arena
    # The light-transport step asks for ~35 zero-initialised buffers (composite targets, scattered gradient rows,
    # masks ...): as torch.zeros each is a ~5 us fill launch, 0.16 ms per step in all.  They are carved out of ONE buffer
    # that is allocated zeroed at the start of the step (a new one per step: results a caller still holds stay intact),
    # sized by the previous step's total; requests beyond it fall back to torch.zeros.
    def _zero_arena_begin(self):
        if not self.zero_arena:
            self._za = None
            return
        need = getattr(self, "_za_used", 0)
        size = (int(need * 1.25) + 4096) // 256 * 256 if need else 0
        self._za = torch.zeros(size, dtype=torch.uint8, device=self.device) if size else None
        self._za_off, self._za_used = 0, 0

    def _z(self, *shape, dtype=torch.float32, device=None):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty(0, dtype=dtype).element_size() + 255) // 256 * 256
        self._za_used = getattr(self, "_za_used", 0) + nbytes
        za = getattr(self, "_za", None)
        if za is None or self._za_off + nbytes > za.numel() or n == 0:
            return torch.zeros(*shape, dtype=dtype, device=self.device)
        v = za[self._za_off: self._za_off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)
        self._za_off += nbytes
        return v

    # ------------------------------------------------------------------ building blocks
    def _march(self, P: Pass, scene, rays_o, rays_d, em_modes, mask_density, sdf, prelude=None, viewdirs=None, between=None):
        """count -> plan -> (host reads the plan header) -> fill.  ``viewdirs``: the rays' view directions -- read under cfg
        neus_alpha "grad" (esrnerf.py:197-200: every march of the renderer extrapolates its section SDFs along them;
        the secondary rays' view directions are their own directions, esrnerf.py:575-591).  ``prelude()``: work that does not depend on the march
        (weight packing, zeroing the gradient buffer), enqueued on a side stream while the host waits; the returned
        event (``P.e_pre``) must be waited for by the main stream before the first consumer.  ``between()``: work for the
        MAIN stream that does not need the march's result, enqueued behind the plan's copy: the device runs it while the host
        reads the plan and enqueues the fill (otherwise it idles for the host's wake-up + first launches, ~0.1 ms)."""
        L, s = self.L, self._s()
        n = rays_o.shape[0]
        P.n_rays = n
        cnt3 = torch.empty(n, dtype=torch.int32, device=self.device)
        off3 = torch.empty(n, dtype=torch.int32, device=self.device)
        last = torch.empty(n, dtype=torch.float32, device=self.device)
        stats = torch.empty(n * 3, dtype=torch.int32, device=self.device)
        sp = C.byref(scene)
        self._run("plan_begin", L.esr_fine_plan_begin, _lib.ptr(self.plan_dev), s)
        ga = self.neus_grad
        if ga:
            if viewdirs is None:
                raise RuntimeError("neus_alpha='grad' marches need the rays' view directions")
            viewdirs = viewdirs.contiguous()
            # (no march cache in this mode: the gradient taps are not recorded; fill and backward walk again)
            self._run(f"march_count[{P.name}]", L.esr_fine_march_count_ga, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                      _lib.ptr(viewdirs), _lib.ptr(mask_density), _lib.ptr(sdf), n, _lib.ptr(cnt3), _lib.ptr(last),
                      _lib.ptr(stats), _lib.ptr(self.plan_dev), s)
            P.march = None
            P.ga = (viewdirs, mask_density, sdf)
        else:
            # march cache: the count pass records every mask-cache survivor; fill copies, the backward starts at its scan
            need = int(L.esr_fine_march_cache_floats(sp, n))
            if getattr(P, "cache", None) is None or P.cache.numel() < need:
                P.cache = torch.empty(need, dtype=torch.float32, device=self.device)
            self._run(f"march_count[{P.name}]", L.esr_fine_march_count_cached, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                      _lib.ptr(mask_density), _lib.ptr(sdf), n, _lib.ptr(cnt3), _lib.ptr(last), _lib.ptr(stats),
                      _lib.ptr(self.plan_dev), _lib.ptr(P.cache), s)
            P.march = (stats, last, P.cache)
        # the counts the host waits for first (a many-workgroup sum), their copy, THEN the one-workgroup scan of the offsets:
        # it runs while the host reads the header and enqueues (17 / 48 us off the path to the read-back)
        self._run("plan_totals", L.esr_fine_plan_totals, _lib.ptr(cnt3), _lib.ptr(em_modes), _lib.ptr(stats), n,
                  _lib.ptr(self.plan_dev), s)
        self.plan_host.copy_(self.plan_dev, non_blocking=True)
        P.e_pre = None
        landed = torch.cuda.Event()
        landed.record()
        self._run("plan", L.esr_fine_plan_offsets, _lib.ptr(cnt3), _lib.ptr(em_modes), n, _lib.ptr(off3), _lib.ptr(self.plan_dev), s)
        if between is not None:
            between()
        if prelude is not None:
            side = self._side_stream(0)
            side.wait_event(landed)
            with torch.cuda.stream(side):
                prelude()
                P.e_pre = torch.cuda.Event()
                P.e_pre.record(side)
        pre_rec = P.bufs.get("rec_ray")       # padding lanes carry ray -1: filled before the wait (fine_engine.forward)
        if pre_rec is not None:
            pre_rec.fill_(-1)
        landed.synchronize()
        n_on, n_off, _, _, m0, m1, m2, overflow = [int(v) for v in self.plan_host.tolist()]
        tiles_on = (n_on + 31) // 32                                 # (esr_fine_plan_totals leaves the tile counts to the host)
        tiles_all = tiles_on + (n_off + 31) // 32
        if overflow & 1:
            self._overflow()
        P.tiles_on, P.tiles_all = tiles_on, tiles_all
        P.counts = dict(m0=m0, m1=m1, m2=m2, m3=n_on + n_off, n_on=n_on, n_off=n_off)
        P.ensure(max(tiles_all, 1))
        rec_ray = P.buf("rec_ray", 1, torch.int32)
        if rec_ray is not pre_rec:                # (the pass's buffers grew: a new, unfilled one)
            rec_ray[: max(tiles_all, 1) * 32].fill_(-1)
        if tiles_all and ga:
            self._run(f"march_fill[{P.name}]", L.esr_fine_march_fill_ga, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                      _lib.ptr(viewdirs), _lib.ptr(mask_density), _lib.ptr(sdf), n, _lib.ptr(off3), _lib.ptr(rec_ray),
                      _lib.ptr(P.buf("rec_step", 1, torch.int32)), _lib.ptr(P.buf("rec_w")),
                      _lib.ptr(P.buf("rec_sdf")), s)
        elif tiles_all:
            self._run(f"march_fill[{P.name}]", L.esr_fine_march_fill_cached, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), n,
                      _lib.ptr(off3), _lib.ptr(stats), _lib.ptr(P.cache), _lib.ptr(rec_ray),
                      _lib.ptr(P.buf("rec_step", 1, torch.int32)), _lib.ptr(P.buf("rec_w")),
                      _lib.ptr(P.buf("rec_sdf")), s)
        P.keep = [rays_o, rays_d, em_modes, cnt3, off3, last, stats]
        return cnt3, off3, last

    def _march_bwd(self, name, P: Pass, sp, rays_o, rays_d, n, off3, dweight, dlast, grad_sdf, dsdf, acc):
        """Backward of one march.  Cached form: the value-tap gradients of the recorded samples go to ``dsdf`` (the
        feature backward folds them into its SDF window).  neus_alpha "grad": a fresh walk that scatters every tap
        straight into ``grad_sdf`` (``dsdf`` is left untouched).  Returns whether ``dsdf`` was written."""
        L, s = self.L, self._s()
        if P.march is None:
            vd, mask_density, sdf = P.ga
            self._run(name, L.esr_fine_march_bwd_ga, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(vd),
                      _lib.ptr(mask_density), _lib.ptr(sdf), n, _lib.ptr(off3), _lib.ptr(dweight), _lib.ptr(dlast),
                      _lib.ptr(grad_sdf), s)
            return False
        st, la, ca = P.march
        self._run(name, L.esr_fine_march_bwd_cached, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), n, _lib.ptr(off3),
                  _lib.ptr(st), _lib.ptr(la), _lib.ptr(ca), _lib.ptr(dweight), _lib.ptr(dlast), _lib.ptr(grad_sdf),
                  _lib.ptr(dsdf) if dsdf is not None else None, acc, s)
        return dsdf is not None

    def _feat_args_records(self, P: Pass, rays_o, rays_d, viewdirs, sdf, color_on, color_off):
        fa = _lib.EsrFeatArgs()
        fa.rays_o, fa.rays_d, fa.viewdirs = rays_o.data_ptr(), rays_d.data_ptr(), viewdirs.data_ptr()
        fa.rec_ray, fa.rec_step = P.bufs["rec_ray"].data_ptr(), P.bufs["rec_step"].data_ptr()
        fa.rec_sdf = P.bufs["rec_sdf"].data_ptr()
        fa.sdf = sdf.data_ptr()
        for g in range(3):
            fa.color_on[g] = color_on[g].data_ptr() if color_on[g] is not None else None
            fa.color_off[g] = color_off[g].data_ptr() if color_off[g] is not None else None
        fa.tiles_on, fa.tiles_all = P.tiles_on, P.tiles_all
        P.fa = fa
        P.keep += [viewdirs]
        return fa

    def _feat_args_points(self, P: Pass, pts, vd, sdfv, sdf, colors):
        n = pts.shape[0]
        tiles = (n + 31) // 32
        P.tiles_on, P.tiles_all = 0, tiles
        P.ensure(tiles)
        fa = _lib.EsrFeatArgs()
        fa.pts, fa.pt_viewdirs, fa.pt_sdf, fa.n_pts = pts.data_ptr(), vd.data_ptr(), sdfv.data_ptr(), n
        fa.sdf = sdf.data_ptr()
        for g in range(3):
            fa.color_on[g] = None
            fa.color_off[g] = colors[g].data_ptr() if colors[g] is not None else None
        fa.tiles_on, fa.tiles_all = 0, tiles
        P.fa = fa
        P.keep = [pts, vd, sdfv]
        return fa

    def _features(self, P: Pass, scene):
        if P.tiles_all:
            self._run(f"feat_fwd[{P.name}]", self.L.esr_fine_feat_fwd, C.byref(scene), C.byref(P.fa),
                      _lib.ptr(P.buf("X", X_ROWS)), _lib.ptr(P.buf("gnorm", 4)), self._s())

    def _net_fwd(self, P: Pass, net, kind, crow, t0, t1, save=True):
        hid, nh, zrows = ((192, 3, 4) if kind == KIND_RADIANCE else (192, 1, 4) if kind == KIND_TONEMAP
                          else (128, 3, 8) if kind == KIND_BRDF else (128, 3, 4))
        H = [P.buf(f"{net}.H{l}", hid) for l in range(nh)]
        M = [P.buf(f"{net}.M{l}", hid // 32 // 2 * 2, torch.int32) for l in range(nh)]
        z = P.buf(f"{net}.z", zrows)
        x = P.bufs["Xt"] if kind == KIND_TONEMAP else P.bufs["X"]
        if t1 > t0:
            # tone mapper: masks only, its weight gradient recomputes the hidden layer (tone_wgrad.hip, f32 and bf16 operands)
            mode = 0 if not save else 2 if kind == KIND_TONEMAP else 1
            if kind == KIND_RADIANCE and self.split_fwd and net in self.packed_split:
                # f32 engine: the radiance forward's products on the 16-bit matrix cores, fp32 results (csrc/mlp_split.hip)
                self._run(f"mlp_fwd({net})[{P.name}]", self.L.esr_mlp_fwd_split, kind, _lib.ptr(self.packed[net]),
                          _lib.ptr(self.packed_split[net]), _lib.ptr(x), t0, t1, _lib.ptr_array(H), _lib.ptr_array(M), mode,
                          crow, _lib.ptr(z), self._s())
            else:
                self._run(f"mlp_fwd({net})[{P.name}]", self.mlp_fwd, kind, _lib.ptr(self.packed[net]),
                          _lib.ptr(x), t0, t1, _lib.ptr_array(H), _lib.ptr_array(M), mode, crow,
                          _lib.ptr(z), self._s())
        return z

    def _net_bwd(self, P: Pass, net, kind, crow, t0, t1, dz, gw, gb):
        """dgrad + wgrad of one net over tiles [t0,t1); returns its dX buffer."""
        hid, nh = ((192, 3) if kind == KIND_RADIANCE else (192, 1) if kind == KIND_TONEMAP else (128, 3))
        H = [P.bufs[f"{net}.H{l}"] for l in range(nh)]
        M = [P.bufs[f"{net}.M{l}"] for l in range(nh)]
        dZ = [P.buf(f"{net}.dZ{l}", hid) for l in range(nh)]
        dX = P.buf(f"{net}.dX", DX_ROWS)
        x = P.bufs["Xt"] if kind == KIND_TONEMAP else P.bufs["X"]
        if t1 > t0:
            s = self._s()
            recompute = kind == KIND_TONEMAP
            amax = None
            if kind in self.split_kinds_bwd and self.split_fwd and self.split_bwd and net in self.packed_split:
                # (radiance, BRDF, emission nets.)  max |dz| of this net and pass, left behind by the input-gradient kernel:
                # the scale of the split-fp16 weight-gradient job (esr_wgrad_job_t::amax)
                amax = self._z(1) if (self.split_wgrad and self._wgrad_jobs is not None) else None
                if recompute and self.split_tone_wgrad:
                    amax = self._z(1)                      # max |dzt|: the scale of the tone mapper's split weight gradients
                self._run(f"mlp_dgrad({net})[{P.name}]", self.L.esr_mlp_dgrad_split, kind, _lib.ptr(self.packed_split[net]),
                          _lib.ptr(dz), t0, t1, _lib.ptr_array(M), _lib.ptr_array([None] * nh if recompute else dZ), _lib.ptr(dX),
                          _lib.ptr(amax) if amax is not None else None, s)
            else:
                self._run(f"mlp_dgrad({net})[{P.name}]", self.mlp_dgrad, kind, _lib.ptr(self.packed[net]),
                          _lib.ptr(dz), t0, t1, _lib.ptr_array(M), _lib.ptr_array([None] * nh if recompute else dZ), _lib.ptr(dX), s)
            if recompute:
                (w0, w1), (b0, _) = self._raw[net]

                def tone_wgrad(amax_t=amax):
                    if self.split_tone_wgrad and amax_t is not None:       # (the scale source comes from the split input-gradient kernel)
                        self._run(f"tone_wgrad[{P.name}]", self.L.esr_tone_wgrad_recompute_split, _lib.ptr(x), _lib.ptr(dz),
                                  _lib.ptr(w0.detach()), _lib.ptr(b0.detach()), _lib.ptr(w1.detach()), _lib.ptr(amax_t), t0, t1,
                                  _lib.ptr(gw[0]), _lib.ptr(gb[0]), _lib.ptr(gw[1]), _lib.ptr(gb[1]), _lib.ptr(self.tone_scratch),
                                  C.c_int64(self.tone_scratch.numel()), self._s())
                        amax_t.record_stream(torch.cuda.current_stream(self.device))
                        return
                    self._run(f"tone_wgrad[{P.name}]",
                              self.L.esr_tone_wgrad_recompute_bf16 if self.bf16 else self.L.esr_tone_wgrad_recompute, _lib.ptr(x), _lib.ptr(dz),
                              _lib.ptr(w0.detach()), _lib.ptr(b0.detach()), _lib.ptr(w1.detach()), t0, t1, _lib.ptr(gw[0]),
                              _lib.ptr(gb[0]), _lib.ptr(gw[1]), _lib.ptr(gb[1]), _lib.ptr(self.tone_scratch),
                              C.c_int64(self.tone_scratch.numel()), self._s())
                if self._wgrad_jobs is not None:
                    self._wgrad_extra.append((tone_wgrad, dz))       # with the batched weight gradients, at the end
                else:
                    tone_wgrad()
            elif self._wgrad_jobs is not None:
                # inside lts_backward: the weight gradients of EVERY net and pass of the step go out as ONE batched call
                # at the end (esr_mlp_wgrad_batch: layers of the same kernel shape share a launch -- eleven net calls
                # with ~100-135 us of fixed cost each become four launch groups)
                self._wgrad_jobs.append((f"{net}[{P.name}]", kind, x, crow, H, dZ, dz, t0, t1, gw, gb, amax))
            else:
                self._run(f"mlp_wgrad({net})[{P.name}]", self.mlp_wgrad, kind, _lib.ptr(x), crow,
                          _lib.ptr_array(H), _lib.ptr_array(dZ), _lib.ptr(dz), t0, t1, _lib.ptr_array(gw),
                          _lib.ptr_array(gb), _lib.ptr(self.wgrad_scratch), C.c_int64(self.wgrad_scratch.numel()), self._s())
        return dX

    def _act(self, P, zname, out, rows, n_ch, act, bwd_g=None, tiles=None):
        tiles = P.tiles_all if tiles is None else tiles
        o = P.buf(out, rows)
        if tiles:
            if bwd_g is None:
                self._run("act_fwd", self.L.esr_act_fwd, _lib.ptr(P.bufs[zname]), tiles, rows, n_ch, act, _lib.ptr(o), self._s())
            else:
                self._run("act_bwd", self.L.esr_act_bwd, _lib.ptr(P.bufs[zname]), _lib.ptr(bwd_g), tiles, rows, n_ch, act,
                          _lib.ptr(o), self._s())
        return o

    def _act_batch(self, name, jobs):
        """One launch for up to four activation jobs (esr_act_batch).  job: dict(P, z, out, rows, n_ch, act[, tiles][, bwd_g]
        [, src, inv][, pt1, ex]) -- ``bwd_g``: tile-major upstream gradient; ``src`` [n, c] row-major gradient rows
        (through ``inv``: slot -> row, or identity); ``ex``: list of ([P, c] tensor, first row) added at the slots
        ``pt1`` marks.  Returns the output buffers."""
        arr = (_lib.EsrActJob * len(jobs))()
        outs, keep, n = [], [], 0
        for jd in jobs:
            P = jd["P"]
            tiles = P.tiles_all if jd.get("tiles") is None else jd["tiles"]
            o = P.buf(jd["out"], jd["rows"])
            outs.append(o)
            if not tiles:
                continue
            jb = arr[n]
            n += 1
            jb.z, jb.out = P.bufs[jd["z"]].data_ptr(), o.data_ptr()
            jb.tiles, jb.rows, jb.n_ch, jb.act = tiles, jd["rows"], jd["n_ch"], jd["act"]
            bwd = any(k in jd for k in ("bwd_g", "src", "ex"))
            jb.bwd = 1 if bwd else 0
            if jd.get("bwd_g") is not None:
                jb.g_tile = jd["bwd_g"].data_ptr()
            if jd.get("src") is not None:
                src = jd["src"].contiguous()
                keep.append(src)
                jb.src, jb.src_c, jb.n_src = src.data_ptr(), src.shape[1], src.shape[0]
                if jd.get("inv") is not None:
                    jb.inv = jd["inv"].data_ptr()
            if jd.get("ex"):
                jb.pt1 = jd["pt1"].data_ptr()
                for e, (t, col0) in enumerate(jd["ex"]):
                    t = t.contiguous()
                    keep.append(t)
                    jb.ex[e], jb.ex_c[e], jb.ex_col0[e] = t.data_ptr(), (t.shape[1] if t.dim() > 1 else 1), col0
        if n:
            self._run(name, self.L.esr_act_batch, arr, n, self._s())
        return outs

    def _gather_batch(self, jobs):
        """One launch for up to four esr_lts_gather_rows jobs: (src, tile_rows, stride, col0, n_ch, perm, n) -> outputs."""
        arr = (_lib.EsrGatherJob * len(jobs))()
        outs = []
        for jb, (src, tile_rows, stride, col0, n_ch, perm, n) in zip(arr, jobs):
            out = torch.empty(n, n_ch, device=self.device)
            outs.append(out)
            jb.src, jb.tile_rows, jb.row_stride, jb.col0, jb.n_ch = src.data_ptr(), tile_rows, stride, col0, n_ch
            jb.perm = perm.data_ptr() if perm is not None else None
            jb.n, jb.out = n, out.data_ptr()
        self._run("lts_gather_rows", self.L.esr_lts_gather_rows_batch, arr, len(jobs), self._s())
        return outs

    def _feat_bwd(self, P: Pass, scene, sources, grad_sdf, dsdf_extra=None, dsdf_out=None, grad4=None, grad4_mode=0):
        if not P.tiles_all:
            return
        src = (_lib.EsrFeatBwdSrc * len(sources))()
        for i, (dX, gon, goff, t0, t1) in enumerate(sources):
            src[i].dX = dX.data_ptr()
            src[i].grad_color_on = gon.data_ptr() if gon is not None else None
            src[i].grad_color_off = goff.data_ptr() if goff is not None else None
            src[i].t0, src[i].t1 = t0, t1
        self._run(f"feat_bwd[{P.name}]", self.L.esr_fine_feat_bwd, C.byref(scene), C.byref(P.fa),
                  _lib.ptr(P.bufs["X"]), _lib.ptr(P.bufs["gnorm"]), src, len(sources), _lib.ptr(dsdf_extra),
                  _lib.ptr(grad_sdf), _lib.ptr(dsdf_out), _lib.ptr(grad4), grad4_mode, self._s())

    def _ref_order(self, P0: Pass, cnt3, off3):
        """perm[k] = compact (tile-order) index of the k-th surviving sample in the reference's ray-sorted
        order; also the int64 ray id of every compact slot (one fused launch after the cumsum)."""
        dev = self.device
        T = P0.tiles_all
        m3 = P0.counts["n_on"] + P0.counts["n_off"]
        csum = torch.cumsum(cnt3, 0, dtype=torch.int64)
        perm = torch.empty(m3, dtype=torch.long, device=dev)
        rec_ray = torch.empty(T * 32, dtype=torch.long, device=dev)
        self.inv_order = torch.empty(T * 32, dtype=torch.int32, device=dev)      # slot -> rank in reference order (-1: padding)
        self._run("lts_ref_order", self.L.esr_lts_ref_order_inv, _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(cnt3), _lib.ptr(off3),
                  _lib.ptr(csum), T * 32, _lib.ptr(perm), _lib.ptr(rec_ray), _lib.ptr(self.inv_order), self._s())
        P0.keep += [csum]
        return perm, rec_ray

    def _gather_rows(self, src, tile_rows, stride, col0, n_ch, perm, n):
        """out[k, c] = src(row perm[k] (or k), column col0 + c): from a tile-major buffer (tile_rows > 0) or a row-major
        one -- one launch instead of permute-copy + gather + slice-copy."""
        out = torch.empty(n, n_ch, device=self.device)
        self._run("lts_gather_rows", self.L.esr_lts_gather_rows, _lib.ptr(src), tile_rows, stride, col0, n_ch,
                  _lib.ptr(perm), n, _lib.ptr(out), self._s())
        return out

    # ------------------------------------------------------------------ image rendering
    @torch.no_grad()
    def evaluate(self, scene, scene2, rays_o, rays_d, viewdirs, grids, envmap, pos_rt, far, em_mode, render_pbr, chunk_sz,
                 num_2ndrays, draws=None):
        """``_evaluate_lts`` + the split kernels' range fallback: one more run on the f32 MFMA kernels, with the same
        scattering draws, when a split launch raised the flag."""
        args = (scene, scene2, rays_o, rays_d, viewdirs, grids, envmap, pos_rt, far, em_mode, render_pbr, chunk_sz, num_2ndrays)
        out = self._evaluate_lts(*args, draws)
        if self.range_hit():
            with self.f32_only():
                out = self._evaluate_lts(*args, self._eval_draws)
        return out

    def _evaluate_lts(self, scene, scene2, rays_o, rays_d, viewdirs, grids, envmap, pos_rt, far, em_mode, render_pbr, chunk_sz,
                      num_2ndrays, draws=None):
        """``ESRNeRF.forward_evaluate`` (esrnerf.py:853-1297), forward only: the 12 image keys of the fine renderer,
        the composited material heads (lin/emit, lin/basecolor, lin/roughness, lin/metallic) and, with ``render_pbr``,
        the light-transport decomposition of EVERY surviving sample (lin/env_dir, lin/env_indir, lin/env_effects,
        lin/emit_(in)dir, lin/emit_effects) evaluated in chunks of ``chunk_sz`` samples x ``num_2ndrays`` secondary rays.
        grids: sdf, off, emo, brdf, emit, mask.  draws: optional list of [chunk, R, 3] standard-normal tensors."""
        L, s, dev = self.L, self._s(), self.device
        sdf, offg, emog, brdfg, emitg = grids["sdf"], grids["off"], grids["emo"], grids["brdf"], grids["emit"]
        n = rays_o.shape[0]
        P0 = self.prim
        self._eval_draws = []                     # the chunks' scattering draws (replayed by the range fallback)
        self._range_event = None
        cnt3, off3, last = self._march(P0, scene, rays_o, rays_d, torch.zeros(n, dtype=torch.int64, device=dev),
                                       grids["mask"], sdf, viewdirs=viewdirs)
        T, m3 = P0.tiles_all, P0.counts["m3"]
        z3 = lambda: torch.zeros(n, 3, dtype=torch.float32, device=dev)
        out = {f"{sp_}/{v}_rgb": z3() for v in ("off", "on", "emo") for sp_ in ("srgb", "lin")}
        out.update({"lin/emit": z3(), "lin/basecolor": z3(), "etc/normal": z3()})
        rm3, depth3 = z3(), z3()
        depth = torch.zeros(n, dtype=torch.float32, device=dev)
        disp = torch.empty(n, dtype=torch.float32, device=dev)
        pbr_keys = ("lin/env_dir", "lin/env_indir", "lin/env_effects", "lin/emit_(in)dir", "lin/emit_effects")
        if render_pbr:
            out.update({k: z3() for k in pbr_keys})
        sp = C.byref(scene)
        comp = lambda src, rows, dst, name: self._run(f"composite3_fwd({name})", L.esr_composite3_fwd, src, rows,
                                                      _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(dst), s)
        if T:
            self._feat_args_records(P0, rays_o, rays_d, viewdirs, sdf, (offg, emog, brdfg), (offg, emog, brdfg))
            self._features(P0, scene)
            self._net_fwd(P0, "off", KIND_RADIANCE, 0, 0, T, save=False)
            self._net_fwd(P0, "emo", KIND_RADIANCE, 88, 0, T, save=False)
            self._net_fwd(P0, "brdf", KIND_BRDF, 96, 0, T, save=False)
            for name, za, zb, ton in (("off", "off.z", "emo.z", 0), ("emo", "emo.z", "emo.z", 0), ("on", "off.z", "emo.z", T)):
                self._run("tone_in_fwd", L.esr_fine_tone_in_fwd, _lib.ptr(P0.bufs[za]), _lib.ptr(P0.bufs[zb]), ton, T,
                          _lib.ptr(P0.buf("lin", 4)), _lib.ptr(P0.buf("Xt", XT_ROWS)), s)
                self._net_fwd(P0, "tone", KIND_TONEMAP, 0, 0, T, save=False)
                self._run("composite_fwd", L.esr_fine_composite_fwd, _lib.ptr(P0.bufs["tone.z"]), _lib.ptr(P0.bufs["lin"]),
                          _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(P0.buf("rgb", 4)),
                          _lib.ptr(out[f"srgb/{name}_rgb"]), _lib.ptr(out[f"lin/{name}_rgb"]), s)
            aux = torch.empty(T * 8 * 32, dtype=torch.float32, device=dev)
            rt = (C.c_float * 9)(*[float(v) for v in pos_rt.detach().cpu().reshape(-1).tolist()])
            self._run("eval_aux", L.esr_eval_aux, _lib.ptr(P0.bufs["X"]), X_ROWS, 40, 36, 32, _lib.ptr(P0.bufs["rec_ray"]),
                      _lib.ptr(P0.bufs["rec_step"]), T, rt, C.c_float(scene.stepdist), _lib.ptr(aux), s)
            comp(_lib.ptr(aux), 8, out["etc/normal"], "normal")
            comp(C.c_void_p(aux.data_ptr() + 4 * 32 * 4), 8, depth3, "depth")
            # material heads; the emission head reads emit_color, which may be a frozen copy distinct from emo_color
            self._act(P0, "brdf.z", "brdf.a", 8, 5, ACT_SIGMOID)
            comp(_lib.ptr(P0.bufs["brdf.a"]), 8, out["lin/basecolor"], "basecolor")
            comp(C.c_void_p(P0.bufs["brdf.a"].data_ptr() + 3 * 32 * 4), 8, rm3, "rough/metal")
            if emitg.data_ptr() == emog.data_ptr():
                self._net_fwd(P0, "emit", KIND_EMIT, 88, 0, T, save=False)
            else:
                keep_x, keep_g = P0.bufs["X"], P0.bufs["gnorm"]
                P0.bufs["X"], P0.bufs["gnorm"] = P0.buf("X.emit", X_ROWS), P0.buf("gnorm.emit", 4)
                self._feat_args_records(P0, rays_o, rays_d, viewdirs, sdf, (emitg, None, None), (emitg, None, None))
                self._features(P0, scene)
                self._net_fwd(P0, "emit", KIND_EMIT, 0, 0, T, save=False)
                P0.bufs["X"], P0.bufs["gnorm"] = keep_x, keep_g
            self._act(P0, "emit.z", "emit.a", 4, 3, ACT_SOFTPLUS)
            comp(_lib.ptr(P0.bufs["emit.a"]), 4, out["lin/emit"], "emit")
        self._run("eval_disp", L.esr_eval_disp, _lib.ptr(depth3), _lib.ptr(last), C.c_float(far), n, _lib.ptr(depth),
                  _lib.ptr(disp), s)
        out.update({"etc/depth": depth, "etc/disp": disp, "etc/white_bg": last.unsqueeze(-1),
                    "lin/roughness": rm3[:, 0].contiguous(), "lin/metallic": rm3[:, 1].contiguous()})
        if render_pbr and T:
            R = int(num_2ndrays)
            perm, rec_ray = self._ref_order(P0, cnt3, off3)
            eg = torch.empty(T * 32, 4, device=dev)
            self._run("expgrad_fwd", L.esr_expgrad_fwd, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(P0.bufs["rec_ray"]),
                      _lib.ptr(P0.bufs["rec_step"]), None, None, C.c_float(0.0), _lib.ptr(sdf), T * 32, 0, _lib.ptr(eg), s)
            pts_all = torch.empty(T * 32, 3, device=dev)
            self._run("sample_points", L.esr_sample_points, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                      _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_step"]), T * 32, _lib.ptr(pts_all), s)
            brdf_rm, emit_rm = P0.rowmajor("brdf.a"), P0.rowmajor("emit.a")
            per = {k: torch.zeros(T * 32, 3, device=dev) for k in pbr_keys}
            P2 = self.sec
            mus, lam, lobes = envmap["mus"].contiguous(), envmap["lambdas"].reshape(-1).contiguous(), envmap["lobes"].contiguous()
            for ci, idx in enumerate(torch.arange(m3, device=dev).split(int(chunk_sz))):
                jc = perm[idx]
                nc = jc.numel()
                pts_c = pts_all[jc].contiguous()
                view_c = viewdirs[rec_ray[jc]].contiguous()
                normal_c = torch.nn.functional.normalize(eg[jc, 1:4], dim=-1).contiguous()
                base_c, rough_c, metal_c = brdf_rm[jc, 0:3].contiguous(), brdf_rm[jc, 3].contiguous(), brdf_rm[jc, 4].contiguous()
                emit_c = emit_rm[jc, 0:3].contiguous()
                raw = self._scatter_draws(nc, R) if draws is None else draws[ci].to(dev).contiguous()
                self._eval_draws.append(raw)
                raw1 = torch.cat([raw, torch.ones(nc, 1, 3, device=dev)], 1).contiguous()       # slot R: unused second view
                dirs_all = torch.empty(nc, R + 1, 3, device=dev)
                self._run("lts_dirs", L.esr_lts_dirs, _lib.ptr(raw1), _lib.ptr(normal_c), nc, R + 1, _lib.ptr(dirs_all), s)
                o2 = pts_c.repeat_interleave(R, 0).contiguous()
                d2 = dirs_all[:, :R].reshape(nc * R, 3).contiguous()
                _, _, last2 = self._march(P2, scene2, o2, d2, torch.zeros(nc * R, dtype=torch.int64, device=dev),
                                          grids["mask"], sdf, viewdirs=d2)
                T2 = P2.tiles_all
                off_m, emo_m = torch.zeros(nc * R, 3, device=dev), torch.zeros(nc * R, 3, device=dev)
                if T2:
                    self._feat_args_records(P2, o2, d2, d2, sdf, (offg, emog, None), (offg, emog, None))
                    self._features(P2, scene2)
                    self._net_fwd(P2, "off", KIND_RADIANCE, 0, 0, T2, save=False)
                    self._net_fwd(P2, "emo", KIND_RADIANCE, 88, 0, T2, save=False)
                    for nm, dst in (("off", off_m), ("emo", emo_m)):
                        self._act(P2, f"{nm}.z", f"{nm}.a", 4, 3, ACT_SOFTPLUS)
                        self._run("composite3_fwd", L.esr_composite3_fwd, _lib.ptr(P2.bufs[f"{nm}.a"]), 4,
                                  _lib.ptr(P2.bufs["rec_ray"]), _lib.ptr(P2.bufs["rec_w"]), T2, _lib.ptr(dst), s)
                zeros3, zeros1 = torch.zeros(nc * R, 3, device=dev), torch.zeros(nc * R, device=dev)
                zero_e = torch.zeros(nc, 3, device=dev)
                um = torch.zeros(nc, dtype=torch.uint8, device=dev)

                def combine(off_in, last_in, emission_in):
                    a = _lib.EsrLtsArgs()
                    a.n_pts, a.n_rays, a.n_sg, a.pdra_mode = nc, R, mus.shape[0], 0
                    held = dict(base=base_c, rough=rough_c, metal=metal_c, normal=normal_c, view=view_c, dirs=dirs_all,
                                off_m=off_in, emo_m=emo_m, last2=last_in, mus=mus, lambdas=lam, lobes=lobes,
                                emission=emission_in, umask=um)
                    for k, v in held.items():
                        setattr(a, k, v.data_ptr())
                    oh, eh = torch.empty(2 * nc, 3, device=dev), torch.empty(2 * nc, 3, device=dev)
                    self._run("lts_combine_fwd", L.esr_lts_combine_fwd, C.byref(a), _lib.ptr(oh), _lib.ptr(eh), s)
                    return oh[:nc], eh[:nc]             # copy 0 = the camera direction
                env_eff, emit_eff = combine(off_m, last2, emit_c)
                env_dir, emit_ind = combine(zeros3, last2, zero_e)
                env_ind, _ = combine(off_m, zeros1, zero_e)
                for k, v in (("lin/env_dir", env_dir), ("lin/env_indir", env_ind), ("lin/env_effects", env_eff),
                             ("lin/emit_(in)dir", emit_ind), ("lin/emit_effects", emit_eff)):
                    per[k][jc] = v
            for k in pbr_keys:
                comp(_lib.ptr(P0.from_rowmajor("pbr.t", 4, per[k])), 4, out[k], k)
        pick = "off" if int(em_mode) == 0 else "on"
        out["srgb/rgb"], out["lin/rgb"] = out[f"srgb/{pick}_rgb"], out[f"lin/{pick}_rgb"]
        self.range_probe()
        return out

    # ------------------------------------------------------------------ PDRA regrouping queries
    @torch.no_grad()
    def eval_query(self, *args):
        out = self._eval_query(*args)
        if self.range_hit():
            with self.f32_only():
                out = self._eval_query(*args)
        return out

    def _eval_query(self, scene, rays_o, rays_d, viewdirs, mask_density, sdf, emit_grid, what: str):
        """``ESRNeRF.eval_emit`` (what="emit", esrnerf.py:1299-1358: composited emission) or ``eval_esp``
        (what="esp", :1360-1407: composited sample position) per ray, forward only -> [N,3]."""
        L, s, dev = self.L, self._s(), self.device
        n = rays_o.shape[0]
        P0 = self.prim
        self._range_event = None
        self._march(P0, scene, rays_o, rays_d, torch.zeros(n, dtype=torch.int64, device=dev), mask_density, sdf,
                    viewdirs=viewdirs)
        T = P0.tiles_all
        out = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        if T == 0:
            return out
        sp = C.byref(scene)
        if what == "emit":
            self._feat_args_records(P0, rays_o, rays_d, viewdirs, sdf, (emit_grid, None, None), (emit_grid, None, None))
            self._features(P0, scene)
            self._net_fwd(P0, "emit", KIND_EMIT, 0, 0, T, save=False)
            v = self._act(P0, "emit.z", "emit.a", 4, 3, ACT_SOFTPLUS)
        else:
            pts = torch.empty(T * 32, 3, device=dev)
            self._run("sample_points", L.esr_sample_points, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                      _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_step"]), T * 32, _lib.ptr(pts), s)
            v = P0.from_rowmajor("pts.t", 4, pts)
        self._run(f"composite3_fwd({what})", L.esr_composite3_fwd, _lib.ptr(v), 4, _lib.ptr(P0.bufs["rec_ray"]),
                  _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(out), s)
        self.range_probe()
        return out

    # ------------------------------------------------------------------ re-lighting fine-tune (A16)
    def _with_draw_closed(self, fn, *args, **kw):
        """Run a forward that may start a surface-point draw (self._pd); whatever happens, the draw's worker is waited for
        and numpy's generator lock released before control leaves the engine (_PointDraw.close)."""
        self._pd = None
        try:
            return fn(*args, **kw)
        finally:
            pd, self._pd = self._pd, None
            if pd is not None:
                pd.close()

    def finetune_forward(self, scene, scene2, batch, grids, cfg, draws=None):
        return self._with_draw_closed(self._finetune_forward, scene, scene2, batch, grids, cfg, draws)

    def _finetune_forward(self, scene, scene2, batch, grids, cfg, draws=None):
        """``ESRNeRF.forward_finetune`` (esrnerf.py:241-484): the emo net's prediction at ``num_ltspts`` surface
        points (camera + one random direction) against edited emission + reflected emo radiance gathered over
        ``num_2ndrays`` secondary rays.  grids: sdf, emo, brdf, emit (the frozen copy feeding the emission head),
        mask.  Only the emo colour grid and the emo net receive gradients (finetune_backward)."""
        L, s, dev = self.L, self._s(), self.device
        self._range_event = None
        sdf, emog, brdfg, emitg = grids["sdf"], grids["emo"], grids["brdf"], grids["emit"]
        rays_o, rays_d, viewdirs = batch["rays_o"], batch["rays_d"], batch["viewdirs"]
        N = rays_o.shape[0]
        P0, P1, P2 = self.prim, self.pts, self.sec
        cnt3, off3, _ = self._march(P0, scene, rays_o, rays_d, torch.zeros(N, dtype=torch.int64, device=dev),
                                    grids["mask"], sdf, viewdirs=viewdirs)
        T, m3 = P0.tiles_all, P0.counts["m3"]
        if T == 0:
            raise RuntimeError("fine-tune step with no surviving sample (degenerate batch)")
        # the surface-point draw (a full host-side shuffle of range(m3): 1.5 ms at C5) on the worker thread, collected below
        point_draw = self._pd = _PointDraw(m3, min(int(cfg["num_ltspts"]), m3), self._draw_ring) if draws is None else None
        sp = C.byref(scene)
        eg = torch.empty(T * 32, 4, device=dev)
        self._run("expgrad_fwd", L.esr_expgrad_fwd, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(P0.bufs["rec_ray"]),
                  _lib.ptr(P0.bufs["rec_step"]), None, None, C.c_float(0.0), _lib.ptr(sdf), T * 32, 0, _lib.ptr(eg), s)
        pts_all = torch.empty(T * 32, 3, device=dev)
        self._run("sample_points", L.esr_sample_points, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_step"]), T * 32, _lib.ptr(pts_all), s)
        perm, rec_ray = self._ref_order(P0, cnt3, off3)
        idx_ref = (point_draw.result() if draws is None else draws["idx"]).to(dev, non_blocking=True)
        Pn, R = idx_ref.numel(), int(cfg["num_2ndrays"])
        jp = perm[idx_ref]
        ray_p = rec_ray[jp]
        pts_p = pts_all[jp].contiguous()
        view_p = viewdirs[ray_p].contiguous()
        normal_p = torch.nn.functional.normalize(eg[jp, 1:4], dim=-1).contiguous()
        sdf_p = P0.bufs["rec_sdf"][: T * 32][jp].contiguous()
        raw = self._scatter_draws(Pn, R + 1) if draws is None or "dirs" not in draws else draws["dirs"].to(dev).contiguous()
        dirs_all = torch.empty(Pn, R + 1, 3, device=dev)
        self._run("lts_dirs", L.esr_lts_dirs, _lib.ptr(raw), _lib.ptr(normal_p), Pn, R + 1, _lib.ptr(dirs_all), s)
        v_rand = (-dirs_all[:, R]).contiguous()
        # heads at the points: colour group rows 0 / 88 / 96 <- emit_color / emo_color / brdf
        pts2, vd2, sdf2 = torch.cat([pts_p, pts_p]).contiguous(), torch.cat([view_p, v_rand]).contiguous(), \
            torch.cat([sdf_p, sdf_p]).contiguous()
        self._feat_args_points(P1, pts2, vd2, sdf2, sdf, (emitg, emog, brdfg))
        self._features(P1, scene)
        T1 = P1.tiles_all
        self._net_fwd(P1, "emo", KIND_RADIANCE, 88, 0, T1)
        self._net_fwd(P1, "brdf", KIND_BRDF, 96, 0, T1, save=False)
        self._net_fwd(P1, "emit", KIND_EMIT, 0, 0, T1, save=False)
        self._act(P1, "emo.z", "emo.a", 4, 3, ACT_SOFTPLUS)
        self._act(P1, "brdf.z", "brdf.a", 8, 5, ACT_SIGMOID)
        self._act(P1, "emit.z", "emit.a", 4, 3, ACT_SOFTPLUS)
        emo_pt = P1.rowmajor("emo.a")[: 2 * Pn, :3].contiguous()
        brdf_rm = P1.rowmajor("brdf.a")[:Pn]
        base_p, rough_p, metal_p = brdf_rm[:, 0:3].contiguous(), brdf_rm[:, 3].contiguous(), brdf_rm[:, 4].contiguous()
        emis_p = P1.rowmajor("emit.a")[:Pn, :3].contiguous()
        modes_p = batch["em_modes"][ray_p].contiguous()
        inten_p = batch["em_intensities"][ray_p].contiguous()
        cols_p = batch["em_colors"][ray_p].contiguous()
        self._run("emit_edit", L.esr_emit_edit, _lib.ptr(emis_p), _lib.ptr(modes_p), _lib.ptr(inten_p), _lib.ptr(cols_p), Pn, s)
        # incoming emo radiance along the secondary rays (no gradient: esrnerf.py:241 no_grad)
        o2 = pts_p.repeat_interleave(R, 0).contiguous()
        d2 = dirs_all[:, :R].reshape(Pn * R, 3).contiguous()
        _, _, last2 = self._march(P2, scene2, o2, d2, torch.zeros(Pn * R, dtype=torch.int64, device=dev), grids["mask"], sdf,
                                  viewdirs=d2)
        T2 = P2.tiles_all
        emo_m = torch.zeros(Pn * R, 3, device=dev)
        if T2:
            self._feat_args_records(P2, o2, d2, d2, sdf, (None, emog, None), (None, emog, None))
            self._features(P2, scene2)
            self._net_fwd(P2, "emo", KIND_RADIANCE, 88, 0, T2, save=False)
            self._act(P2, "emo.z", "emo.a", 4, 3, ACT_SOFTPLUS)
            self._run("composite3_fwd", L.esr_composite3_fwd, _lib.ptr(P2.bufs["emo.a"]), 4, _lib.ptr(P2.bufs["rec_ray"]),
                      _lib.ptr(P2.bufs["rec_w"]), T2, _lib.ptr(emo_m), s)
        a = _lib.EsrLtsArgs()
        a.n_pts, a.n_rays, a.n_sg, a.pdra_mode = Pn, R, 1, 0
        zero3 = torch.zeros(1, 3, device=dev)
        held = dict(base=base_p, rough=rough_p, metal=metal_p, normal=normal_p, view=view_p, dirs=dirs_all,
                    off_m=torch.zeros(Pn * R, 3, device=dev), emo_m=emo_m, last2=torch.zeros(Pn * R, device=dev),
                    mus=zero3, lambdas=torch.ones(1, device=dev), lobes=torch.ones(1, 3, device=dev), emission=emis_p,
                    umask=torch.zeros(Pn, dtype=torch.uint8, device=dev))
        for k, v in held.items():
            setattr(a, k, v.data_ptr())
        off_hat = torch.empty(2 * Pn, 3, device=dev)          # the off half of the combine is unused here
        emo_hat = torch.empty(2 * Pn, 3, device=dev)
        self._run("lts_combine_fwd", L.esr_lts_combine_fwd, C.byref(a), _lib.ptr(off_hat), _lib.ptr(emo_hat), s)
        ctx = dict(scene=scene, n_pts=Pn, held=held, keep=(pts2, vd2, sdf2, last2), f32_only=not self.bf16 and not self.split_fwd)
        self.last_draws = dict(idx=idx_ref, dirs=raw)
        self.range_probe()
        return ctx, {"lin/pbr/emo": emo_pt, "lin/pbr/emo_hat": emo_hat}

    def finetune_backward(self, ctx, g_emo, grads):
        """g_emo [2*P,3] -> grads["emo"] (colour grid, channels-last), grads["emo_w"], grads["emo_b"]."""
        if ctx.get("f32_only") and self.split_fwd:
            with self.f32_only():
                return self.finetune_backward(ctx, g_emo, grads)
        P1, dev = self.pts, self.device
        T1, Pn = P1.tiles_all, ctx["n_pts"]
        ga = torch.zeros(T1 * 32, 3, device=dev)
        ga[: 2 * Pn] = g_emo
        gt = P1.from_rowmajor("emo.ga", 4, ga)
        dz = self._act(P1, "emo.z", "emo.dz", 4, 3, ACT_SOFTPLUS, bwd_g=gt)
        dX = self._net_bwd(P1, "emo", KIND_RADIANCE, 88, 0, T1, dz, grads["emo_w"], grads["emo_b"])
        self._feat_bwd(P1, ctx["scene"], [(dX, None, grads["emo"], 0, T1)], None)

    # ------------------------------------------------------------------ forward
    def lts_forward(self, scene, scene2, batch, grids, envmap, cfg, draws=None, prelude=None):
        return self._with_draw_closed(self._lts_forward, scene, scene2, batch, grids, envmap, cfg, draws, prelude)

    def _lts_forward(self, scene, scene2, batch, grids, envmap, cfg, draws=None, prelude=None):
        """grids: dict sdf [X,Y,Z], off/emo/brdf [X,Y,Z,6], mask [mx,my,mz].  cfg: num_2ndrays, num_ltspts,
        normal_eps, emit_eps, pdra.  draws (optional): dict idx, dirs, noise_normal, noise_emit -- when
        absent they are drawn exactly where the reference draws them."""
        L, s, dev = self.L, self._s(), self.device
        sdf, offg, emog, brdfg = grids["sdf"], grids["off"], grids["emo"], grids["brdf"]
        self._range_event = None
        self._zero_arena_begin()
        rays_o, rays_d, viewdirs = batch["rays_o"], batch["rays_d"], batch["viewdirs"]
        N = rays_o.shape[0]
        P0 = self.prim
        cnt3, off3, last = self._march(P0, scene, rays_o, rays_d, batch["em_modes"], grids["mask"], sdf, prelude=prelude,
                                       viewdirs=viewdirs)
        T, Ton = P0.tiles_all, P0.tiles_on
        srgb = self._z(N, 3, device=dev)
        lin_m = self._z(N, 3, device=dev)
        emit_m = self._z(N, 3, device=dev)
        ctx = LtsCtx(scene=scene, scene2=scene2, batch=batch, perm=None, jp=None, n_pts=0,
                     n_2nd=int(cfg["num_2ndrays"]), pdra=bool(cfg["pdra"]))
        ctx.f32_only = not self.bf16 and not self.split_fwd
        ctx.t.update(cnt3=cnt3, off3=off3, last=last, grids=grids, envmap=envmap)
        ctx.eps = dict(normal=float(cfg["normal_eps"]), emit=float(cfg["emit_eps"]))
        m3 = P0.counts["m3"]
        if T == 0:
            raise RuntimeError("LTS step with no surviving sample (degenerate batch)")
        sp = C.byref(scene)
        self._feat_args_records(P0, rays_o, rays_d, viewdirs, sdf, (offg, emog, brdfg), (offg, emog, brdfg))
        self._features(P0, scene)
        # (handing the draw to the library's worker thread is ~7 us of host time; the draw itself is 0.2-0.4 ms against
        # ~0.6 ms of primary-pass work queued in front of its consumer)
        point_draw = self._pd = _PointDraw(m3, min(int(cfg["num_ltspts"]), m3), self._draw_ring) if draws is None else None
        # exact normals (+ positions) of every surviving sample
        eg = torch.empty(T * 32, 4, device=dev)
        self._run("expgrad_fwd", L.esr_expgrad_fwd, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(P0.bufs["rec_ray"]),
                  _lib.ptr(P0.bufs["rec_step"]), None, None, C.c_float(0.0), _lib.ptr(sdf), T * 32, 0, _lib.ptr(eg), s)
        pts_all = torch.empty(T * 32, 3, device=dev)
        self._run("sample_points", L.esr_sample_points, sp, _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_step"]), T * 32, _lib.ptr(pts_all), s)
        if P0.e_pre is not None:                  # packed weights (and the zeroed gradient buffer) from the side stream
            torch.cuda.current_stream(dev).wait_event(P0.e_pre)
        # radiance heads: off on every tile, emo on the on-tiles (both carry gradients, esrnerf.py:751-757)
        self._net_fwd(P0, "off", KIND_RADIANCE, 0, 0, T)
        self._net_fwd(P0, "emo", KIND_RADIANCE, 88, 0, Ton)
        self._run("tone_in_fwd", L.esr_fine_tone_in_fwd, _lib.ptr(P0.bufs["off.z"]), _lib.ptr(P0.bufs["emo.z"]), Ton, T,
                  _lib.ptr(P0.buf("lin", 4)), _lib.ptr(P0.buf("Xt", XT_ROWS)), s)
        self._net_fwd(P0, "tone", KIND_TONEMAP, 0, 0, T)
        self._run("composite_fwd", L.esr_fine_composite_fwd, _lib.ptr(P0.bufs["tone.z"]), _lib.ptr(P0.bufs["lin"]),
                  _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(P0.buf("rgb", 4)),
                  _lib.ptr(srgb), _lib.ptr(lin_m), s)
        # material heads on every sample
        self._net_fwd(P0, "brdf", KIND_BRDF, 96, 0, T)
        self._net_fwd(P0, "emit", KIND_EMIT, 88, 0, T)
        self._act_batch("act_fwd", [dict(P=P0, z="brdf.z", out="brdf.a", rows=8, n_ch=5, act=ACT_SIGMOID),
                                    dict(P=P0, z="emit.z", out="emit.a", rows=4, n_ch=3, act=ACT_SOFTPLUS)])
        self._run("composite3_fwd(emit)", L.esr_composite3_fwd, _lib.ptr(P0.bufs["emit.a"]), 4,
                  _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(emit_m), s)

        # ---- compact order <-> the reference's ray-sorted order
        perm, rec_ray = self._ref_order(P0, cnt3, off3)
        ctx.perm = perm

        # ---- random draws in the reference's order (scattering directions, then the two perturbations), then the
        # perturbed re-evaluations BEFORE the light-transport segment: that segment starts with ~45 small gathers whose
        # enqueueing is host-bound (~15 us each against ~5 us on the device); with the large kernels of this block queued
        # in front of them the device stays busy meanwhile (tools/trace_step.py: 0.5 ms idle per step otherwise)
        Pn_, R_ = min(int(cfg["num_ltspts"]), m3) if draws is None else int(draws["idx"].numel()), ctx.n_2nd
        raw = self._scatter_draws(Pn_, R_ + 1) if draws is None or "dirs" not in draws else draws["dirs"].to(dev).contiguous()
        # perturbed re-evaluations (esrnerf.py:807-830)
        nn_ = torch.randn(m3, 3, device=dev) if draws is None else draws["noise_normal"].to(dev)
        ne_ = torch.randn(m3, 3, device=dev) if draws is None else draws["noise_emit"].to(dev)
        noise_n = self._z(T * 32, 3, device=dev)
        pts_e = torch.empty(m3, 3, device=dev)
        nn_, ne_ = nn_.contiguous(), ne_.contiguous()
        self._run("lts_perturb", L.esr_lts_perturb, _lib.ptr(pts_all), _lib.ptr(perm), _lib.ptr(nn_), _lib.ptr(ne_),
                  C.c_float(ctx.eps["emit"]), m3, _lib.ptr(noise_n), _lib.ptr(pts_e), s)
        P0.keep += [nn_, ne_]
        eg_eps = torch.empty(T * 32, 4, device=dev)
        self._run("expgrad_fwd(eps)", L.esr_expgrad_fwd, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(P0.bufs["rec_ray"]),
                  _lib.ptr(P0.bufs["rec_step"]), None, _lib.ptr(noise_n), C.c_float(ctx.eps["normal"]), _lib.ptr(sdf),
                  T * 32, 0, _lib.ptr(eg_eps), s)
        ctx.t.update(noise_n=noise_n)
        # emit_eps / brdf_eps: forward only (explicit points in reference order)
        P3 = self.epsp
        sv = torch.empty(m3, 4, device=dev)
        sdf_e = torch.empty(m3, device=dev)
        vd_e = self._z(m3, 3, device=dev)
        eps_grads = bool(cfg.get("eps_grads", True))      # keep activations for d/d(emit_eps, brdf_eps)
        eps_out = {}

        def eps_pass():
            self._run("expgrad_fwd(pts)", L.esr_expgrad_fwd, sp, None, None, None, None, _lib.ptr(pts_e), None, C.c_float(0.0),
                      _lib.ptr(sdf), m3, 1, _lib.ptr(sv), self._s())
            sdf_e.copy_(sv[:, 0])
            self._feat_args_points(P3, pts_e, vd_e, sdf_e, sdf, (None, emog, brdfg))
            self._features(P3, scene)
            T3 = P3.tiles_all
            self._net_fwd(P3, "emit", KIND_EMIT, 88, 0, T3, save=eps_grads)
            self._net_fwd(P3, "brdf", KIND_BRDF, 96, 0, T3, save=eps_grads)
            self._act_batch("act_fwd", [dict(P=P3, z="emit.z", out="emit.a", rows=4, n_ch=3, act=ACT_SOFTPLUS),
                                        dict(P=P3, z="brdf.z", out="brdf.a", rows=8, n_ch=5, act=ACT_SIGMOID)])
            eps_out["emit"], eps_out["brdf"] = self._gather_batch([(P3.bufs["emit.a"], 4, 0, 0, 3, None, m3),
                                                                   (P3.bufs["brdf.a"], 8, 0, 0, 5, None, m3)])
        eps_done = None
        if self.overlap_wgrad and self.eps_stream:
            # this pass feeds nothing before the loss: on a stream of its own it runs beside the light-transport segment's
            # gathers and the secondary march (latency-bound kernels) instead of in front of them
            main_, side_ = torch.cuda.current_stream(dev), self._side_stream(1)
            ev_ = torch.cuda.Event()
            ev_.record(main_)
            side_.wait_event(ev_)
            with torch.cuda.stream(side_):
                eps_pass()
                eps_done = torch.cuda.Event()
                eps_done.record(side_)
            for t_ in eps_out.values():
                t_.record_stream(main_)
        else:
            eps_pass()
        emit_eps, brdf_eps = eps_out["emit"], eps_out["brdf"]


        # ---- light-transport segment
        idx_host = point_draw.result() if draws is None else draws["idx"]
        self.last_point_idx = idx_host
        idx_ref = idx_host.to(dev, non_blocking=True)
        Pn, R = idx_ref.numel(), ctx.n_2nd
        ctx.n_pts = Pn
        jp = perm[idx_ref]
        ctx.jp = jp
        # everything the segment needs at the points, one launch (pts2 / sdf2: the point twice; vd2: camera direction |
        # random direction, the second half filled below)
        pts2, vd2, sdf2 = torch.empty(2 * Pn, 3, device=dev), torch.empty(2 * Pn, 3, device=dev), torch.empty(2 * Pn, device=dev)
        normal_p, base_p, emis_p = (torch.empty(Pn, 3, device=dev) for _ in range(3))           # (detached normals)
        rough_p, metal_p = torch.empty(Pn, device=dev), torch.empty(Pn, device=dev)
        umask_p = torch.empty(Pn, dtype=torch.uint8, device=dev)
        um_rays = batch["uncert_masks"]
        if um_rays.dtype not in (torch.bool, torch.uint8):
            um_rays = um_rays.to(torch.uint8)
        um_rays = um_rays.contiguous()
        pt1 = self._z(T * 32, dtype=torch.int32, device=dev)     # slot -> surface point + 1 (the backward's gathers)
        gp = _lib.EsrLtsGather()
        gp.n_pts = Pn
        for k, v in dict(jp=jp, ray64=rec_ray, pts_all=pts_all, eg=eg, rec_sdf=P0.bufs["rec_sdf"], viewdirs=viewdirs,
                         brdf_a=P0.bufs["brdf.a"], emit_a=P0.bufs["emit.a"], umask_rays=um_rays, pts2=pts2, vd2=vd2, sdf2=sdf2,
                         normal=normal_p, base=base_p, rough=rough_p, metal=metal_p, emis=emis_p, umask=umask_p, pt1=pt1).items():
            setattr(gp, k, v.data_ptr())
        self._run("lts_gather_points", L.esr_lts_gather_points, C.byref(gp), s)
        ctx.t.update(pt1=pt1)
        pts_p, view_p, sdf_p = pts2[:Pn], vd2[:Pn], sdf2[:Pn]
        # hemisphere directions + the secondary rays' origins / directions + the random view direction (vd2's second half)
        dirs_all = torch.empty(Pn, R + 1, 3, device=dev)
        o2, d2 = torch.empty(Pn * R, 3, device=dev), torch.empty(Pn * R, 3, device=dev)
        self._run("lts_dirs", L.esr_lts_dirs_rays, _lib.ptr(raw), _lib.ptr(normal_p), _lib.ptr(pts_p), Pn, R + 1,
                  _lib.ptr(dirs_all), _lib.ptr(o2), _lib.ptr(d2), C.c_void_p(vd2.data_ptr() + Pn * 3 * 4), s)
        # (a) radiance predicted by the nets at the points, camera direction and random direction
        P1 = self.pts
        at_pts = {}

        def points_pass():
            self._feat_args_points(P1, pts2, vd2, sdf2, sdf, (offg, emog, None))
            self._features(P1, scene)
            T1 = P1.tiles_all
            self._net_fwd(P1, "off", KIND_RADIANCE, 0, 0, T1)
            self._net_fwd(P1, "emo", KIND_RADIANCE, 88, 0, T1)
            self._act_batch("act_fwd", [dict(P=P1, z="off.z", out="off.a", rows=4, n_ch=3, act=ACT_SOFTPLUS),
                                        dict(P=P1, z="emo.z", out="emo.a", rows=4, n_ch=3, act=ACT_SOFTPLUS)])
            at_pts["off"], at_pts["emo"] = self._gather_batch([(P1.bufs["off.a"], 4, 0, 0, 3, None, 2 * Pn),
                                                               (P1.bufs["emo.a"], 4, 0, 0, 3, None, 2 * Pn)])
        # (b) incoming radiance along the secondary rays; (a) is enqueued behind the march's count + plan, so the device
        # works on it while the host waits for the plan header and enqueues the fill
        P2 = self.sec
        em2 = self._z(Pn * R, dtype=torch.int64, device=dev)
        _, off3_2, last2 = self._march(P2, scene2, o2, d2, em2, grids["mask"], sdf, viewdirs=d2, between=points_pass)
        off_pt, emo_pt = at_pts["off"], at_pts["emo"]
        T2 = P2.tiles_all
        off_m = self._z(Pn * R, 3, device=dev)
        emo_m = self._z(Pn * R, 3, device=dev)
        if T2:
            self._feat_args_records(P2, o2, d2, d2, sdf, (offg, emog, None), (offg, emog, None))
            self._features(P2, scene2)
            self._net_fwd(P2, "off", KIND_RADIANCE, 0, 0, T2)
            self._net_fwd(P2, "emo", KIND_RADIANCE, 88, 0, T2)
            self._act_batch("act_fwd", [dict(P=P2, z="off.z", out="off.a", rows=4, n_ch=3, act=ACT_SOFTPLUS),
                                        dict(P=P2, z="emo.z", out="emo.a", rows=4, n_ch=3, act=ACT_SOFTPLUS)])
            for nm, dst in (("off.a", off_m), ("emo.a", emo_m)):
                self._run("composite3_fwd", L.esr_composite3_fwd, _lib.ptr(P2.bufs[nm]), 4, _lib.ptr(P2.bufs["rec_ray"]),
                          _lib.ptr(P2.bufs["rec_w"]), T2, _lib.ptr(dst), s)
        # (c) rendering equation
        a = _lib.EsrLtsArgs()
        a.n_pts, a.n_rays, a.n_sg, a.pdra_mode = Pn, R, envmap["mus"].shape[0], 1 if ctx.pdra else 0
        lam = envmap["lambdas"].reshape(-1).contiguous()
        held = dict(base=base_p, rough=rough_p, metal=metal_p, normal=normal_p, view=view_p, dirs=dirs_all,
                    off_m=off_m, emo_m=emo_m, last2=last2, mus=envmap["mus"].contiguous(), lambdas=lam,
                    lobes=envmap["lobes"].contiguous(), emission=emis_p, umask=umask_p)
        for k, v in held.items():
            setattr(a, k, v.data_ptr())
        off_hat = torch.empty(2 * Pn, 3, device=dev)
        emo_hat = torch.empty(2 * Pn, 3, device=dev)
        self._run("lts_combine_fwd", L.esr_lts_combine_fwd, C.byref(a), _lib.ptr(off_hat), _lib.ptr(emo_hat), s)
        ctx.t.update(held=held, lts_args=a, off3_2=off3_2, o2=o2, d2=d2, eg=eg, pts_all=pts_all)

        um = batch["uncert_masks"]
        r_normal, r_normal_eps, r_emit, r_brdf = self._gather_batch([
            (eg, 0, 4, 1, 3, perm, m3), (eg_eps, 0, 4, 1, 3, perm, m3),
            (P0.bufs["emit.a"], 4, 0, 0, 3, perm, m3), (P0.bufs["brdf.a"], 8, 0, 0, 5, perm, m3)])
        out = {
            "etc/alphainv_cum": last, "srgb/rgb": srgb, "lin/rgb": lin_m,
            "lin/pbr/off": off_pt, "lin/pbr/off_hat": off_hat, "lin/pbr/emo": emo_pt, "lin/pbr/emo_hat": emo_hat,
            "emit_marched": emit_m,
            "etc/normal": r_normal, "etc/normal_eps": r_normal_eps,
            "etc/emit": r_emit, "etc/emit_eps": emit_eps,
            "etc/brdf": r_brdf, "etc/brdf_eps": brdf_eps,
        }
        ctx.t.update(um=um, pts_e=pts_e, eps_grads=eps_grads, m3=m3, inv=self.inv_order)
        if eps_done is not None:
            torch.cuda.current_stream(dev).wait_event(eps_done)
        self.last_draws = dict(idx=idx_host, dirs=raw, noise_normal=nn_, noise_emit=ne_)
        self.range_probe()                        # behind the forward's last split launch (every stream is joined here)
        return ctx, out

    # ------------------------------------------------------------------ backward
    def lts_backward(self, ctx: LtsCtx, g: Dict[str, Optional[torch.Tensor]], grads, after_grids=None):
        """g: gradients w.r.t. the tensors of lts_forward's dict (None = zero).  grads: zero-initialised
        dict: sdf, off, emo, brdf grids; {off,emo,tone,brdf,emit}_{w,b} lists; mus, lambdas, lobes.
        Order on the main stream: every input-gradient chain and grid scatter, then ``after_grids()`` (the
        data-parallel step exchanges the dense-grid gradients there); the weight-gradient launches run beside it on
        a second stream (or, without ``overlap_wgrad``, after it) and are joined at the end."""
        if ctx.f32_only and self.split_fwd:         # (the range fallback's forward: fine_engine.FineEngine.backward)
            with self.f32_only():
                return self.lts_backward(ctx, g, grads, after_grids)
        main = torch.cuda.current_stream(self.device)
        self._wgrad_jobs, self._wgrad_extra = [], []
        self._wgrad_flushed = False
        self._scatter_done = None
        self.last_wgrad_jobs = []
        try:
            self._lts_backward(ctx, g, grads)
            if self._scatter_done is not None:          # (before the grid gradients' exchange and anything else that reads them)
                main.wait_event(self._scatter_done)
                self._scatter_done = None
            jobs_done = None
            if self._wgrad_jobs or self._wgrad_extra or self._wgrad_flushed:
                if self.overlap_wgrad:          # beside the dense-grid exchange / whatever follows on the main stream
                    side = self._side_stream()
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        if self._wgrad_jobs or self._wgrad_extra:
                            self._launch_wgrad_jobs()
                        jobs_done = torch.cuda.Event()
                        jobs_done.record(side)
                else:
                    if after_grids is not None:
                        after_grids()
                        after_grids = None
                    self._launch_wgrad_jobs()
            if after_grids is not None:
                after_grids()
            if jobs_done is not None:
                main.wait_event(jobs_done)
        finally:
            self._wgrad_jobs = None

    def _on_scatter_stream(self, which, fn):
        """Inside ``_lts_backward``: run the grid scatters ``fn`` (march_bwd + feat_bwd of one pass) on the scatter stream, behind
        everything enqueued on the main stream so far.  Nothing on the main stream reads what they write (atomic sums into the
        grid gradients) before ``lts_backward`` joins the stream.  ``which``: 1 = secondary pass, 2 = primary pass
        (``self.scatter_streamed`` lists the ones that leave the main stream)."""
        if not (self.overlap_wgrad and which in self.scatter_streamed):
            fn()
            return
        main, scat = torch.cuda.current_stream(self.device), self._side_stream(1)
        ev = torch.cuda.Event()
        ev.record(main)
        scat.wait_event(ev)
        with torch.cuda.stream(scat):
            fn()
            self._scatter_done = torch.cuda.Event()
            self._scatter_done.record(scat)

    def _flush_wgrad(self, point):
        """Inside ``_lts_backward``: send the weight-gradient jobs collected so far to the second stream NOW, so that they run
        beside the grid scatters that follow on the main stream (march_bwd + feat_bwd: LDS / L2 atomics, the matrix cores idle)
        instead of after them.  ``point``: 1 = behind the secondary pass's input gradients, 3 = behind the primary pass's
        radiance nets', 2 = behind its material heads' (4: the points' pass, 5: the perturbed heads' -- off by default)."""
        if not (self.overlap_wgrad and point in self.wgrad_early and (self._wgrad_jobs or self._wgrad_extra)):
            return
        main, side = torch.cuda.current_stream(self.device), self._side_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self._launch_wgrad_jobs()
        self._wgrad_jobs, self._wgrad_extra = [], []
        self._wgrad_flushed = True

    def _launch_wgrad_jobs(self):
        jobs = self._wgrad_jobs
        self.last_wgrad_jobs += [(name, t1 - t0) for name, _, _, _, _, _, _, t0, t1, _, _, _ in jobs]    # (net[pass], tiles) of the step
        arr = (_lib.EsrWgradJob * len(jobs))()
        keep = []
        for jb, (_, kind, x, crow, H, dZ, dz, t0, t1, gw, gb, amax) in zip(arr, jobs):
            ptrs = [_lib.ptr_array(H), _lib.ptr_array(dZ), _lib.ptr_array(gw), _lib.ptr_array(gb)]
            keep.append(ptrs)
            if amax is not None:
                jb.amax = amax.data_ptr()
                amax.record_stream(torch.cuda.current_stream(self.device))
            jb.kind, jb.color_row0, jb.t0, jb.t1 = kind, crow, t0, t1
            jb.X, jb.dz = x.data_ptr(), dz.data_ptr()
            jb.H, jb.dZ, jb.gw, jb.gb = (C.addressof(p) for p in ptrs)
            dz.record_stream(torch.cuda.current_stream(self.device))      # may be a transient allocation of the main stream
        if jobs:
            self._run("mlp_wgrad(all)", self.L.esr_mlp_wgrad_batch, arr, len(jobs), 1 if self.bf16 else 0,
                      _lib.ptr(self.wgrad_scratch), C.c_int64(self.wgrad_scratch.numel()), self._s())
        for fn, dz in self._wgrad_extra:
            dz.record_stream(torch.cuda.current_stream(self.device))
            fn()

    def _lts_backward(self, ctx: LtsCtx, g: Dict[str, Optional[torch.Tensor]], grads):
        L, s, dev = self.L, self._s(), self.device
        P0, P1, P2 = self.prim, self.pts, self.sec
        T, Ton = P0.tiles_all, P0.tiles_on
        Pn, R = ctx.n_pts, ctx.n_2nd
        perm, jp = ctx.perm, ctx.jp
        zero = lambda k, shape: g[k].contiguous() if g.get(k) is not None else self._z(shape, device=dev)
        sp, sp2 = C.byref(ctx.scene), C.byref(ctx.scene2)
        b = ctx.batch
        grid_g = grads

        # ---- rendering equation
        g_oh, g_eh = zero("lin/pbr/off_hat", (2 * Pn, 3)), zero("lin/pbr/emo_hat", (2 * Pn, 3))
        z = lambda *sh: self._z(*sh, device=dev)
        d = dict(d_off_m=z(Pn * R, 3), d_emo_m=z(Pn * R, 3), d_last2=z(Pn * R), d_base=z(Pn, 3), d_rough=z(Pn),
                 d_metal=z(Pn), d_emission=z(Pn, 3), d_mus=grads["mus"], d_lambdas=grads["lambdas"].view(-1),
                 d_lobes=grads["lobes"])
        gs = _lib.EsrLtsGrads()
        for k, v in d.items():
            setattr(gs, k, v.data_ptr())
        self._run("lts_combine_bwd", L.esr_lts_combine_bwd, C.byref(ctx.t["lts_args"]), _lib.ptr(g_oh), _lib.ptr(g_eh),
                  C.byref(gs), s)

        # ---- secondary rays
        T2 = P2.tiles_all
        if T2:
            dw2 = P2.buf("dweight")
            for i, (nm, gm) in enumerate((("off", d["d_off_m"]), ("emo", d["d_emo_m"]))):
                da = P2.buf(f"{nm}.da", 4)
                self._run("composite3_bwd", L.esr_composite3_bwd, _lib.ptr(gm), _lib.ptr(P2.bufs[f"{nm}.a"]), 4,
                          _lib.ptr(P2.bufs["rec_ray"]), _lib.ptr(P2.bufs["rec_w"]), T2, 2 if i else 0, _lib.ptr(da),
                          _lib.ptr(dw2), s)      # (second call: fresh dv, accumulated dweight)
            src = []
            dzs = self._act_batch("act_bwd", [dict(P=P2, z=f"{nm}.z", out=f"{nm}.dz", rows=4, n_ch=3, act=ACT_SOFTPLUS,
                                                   bwd_g=P2.bufs[f"{nm}.da"]) for nm in ("off", "emo")])
            for (nm, crow, gon), dz in zip((("off", 0, grads["off"]), ("emo", 88, grads["emo"])), dzs):
                dX = self._net_bwd(P2, nm, KIND_RADIANCE, crow, 0, T2, dz, grads[f"{nm}_w"], grads[f"{nm}_b"])
                src.append((dX, None, gon, 0, T2))
            self._flush_wgrad(1)
            # the secondary march's value-tap gradients of the recorded samples ride on the feature backward's window
            ds2 = P2.buf("dsdf")

            def scatter2():
                wrote = self._march_bwd("march_bwd[secondary]", P2, sp2, ctx.t["o2"], ctx.t["d2"], Pn * R, ctx.t["off3_2"], dw2,
                                        d["d_last2"], grads["sdf"], ds2, 0)
                self._feat_bwd(P2, ctx.scene2, src, grads["sdf"], dsdf_extra=ds2 if wrote else None)
            # the secondary pass's grid scatters (incoherent rays: L2 atomics, ~0.4 ms with the matrix cores idle) on a stream of
            # their own, beside the input-gradient chains of the points' and the primary pass that follow on the main stream
            self._on_scatter_stream(1, scatter2)
        else:
            self._march_bwd("march_bwd[secondary]", P2, sp2, ctx.t["o2"], ctx.t["d2"], Pn * R, ctx.t["off3_2"], z(32),
                            d["d_last2"], grads["sdf"], None, 0)

        # ---- radiance at the points
        T1 = P1.tiles_all
        src = []
        # the gradient rows [2P, 3] of the two heads at the points go straight into the activation's backward (slot k <- row k)
        dzs = self._act_batch("act_bwd", [dict(P=P1, z=f"{nm}.z", out=f"{nm}.dz", rows=4, n_ch=3, act=ACT_SOFTPLUS,
                                               src=zero(key, (2 * Pn, 3))) for nm, key in (("off", "lin/pbr/off"), ("emo", "lin/pbr/emo"))])
        for (nm, crow, gon), dz in zip((("off", 0, grads["off"]), ("emo", 88, grads["emo"])), dzs):
            dX = self._net_bwd(P1, nm, KIND_RADIANCE, crow, 0, T1, dz, grads[f"{nm}_w"], grads[f"{nm}_b"])
            src.append((dX, None, gon, 0, T1))
        dsdf_pts = self._z(T1 * 32, device=dev)
        self._flush_wgrad(4)
        self._feat_bwd(P1, ctx.scene, src, grads["sdf"], dsdf_out=dsdf_pts)

        # ---- primary pass: assemble per-sample head gradients in compact order
        dsdf_extra = self._z(T * 32, device=dev)
        dsdf_extra.index_add_(0, jp, dsdf_pts[:Pn] + dsdf_pts[Pn: 2 * Pn])
        # composites
        g_srgb, g_lin = zero("srgb/rgb", (P0.n_rays, 3)), zero("lin/rgb", (P0.n_rays, 3))
        g_em = zero("emit_marched", (P0.n_rays, 3))
        g_last = zero("etc/alphainv_cum", (P0.n_rays,))
        dweight = P0.buf("dweight")
        self._run("composite_bwd", L.esr_fine_composite_bwd, _lib.ptr(g_srgb), _lib.ptr(g_lin), _lib.ptr(P0.bufs["rgb"]),
                  _lib.ptr(P0.bufs["lin"]), _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, _lib.ptr(dweight),
                  _lib.ptr(P0.buf("tone.dz", 4)), s)
        # d(emission head): the composite's share lands in a FRESH tile-major buffer (mode 2: dv written, dweight
        # accumulated); the reference-order rows of etc/emit and the rendering equation's d_emission at the surface points
        # are gathered inside the activation's backward below (esr_act_batch) -- no index_put / index_add_ / permute-copy
        d_emit_t = P0.buf("emit.da", 4)
        self._run("composite3_bwd(emit)", L.esr_composite3_bwd, _lib.ptr(g_em), _lib.ptr(P0.bufs["emit.a"]), 4,
                  _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_w"]), T, 2, _lib.ptr(d_emit_t), _lib.ptr(dweight), s)
        # tonemapper -> radiance heads
        dXt = self._net_bwd(P0, "tone", KIND_TONEMAP, 0, 0, T, P0.bufs["tone.dz"], grads["tone_w"], grads["tone_b"])
        self._run("lts_tone_in_bwd", L.esr_lts_tone_in_bwd, _lib.ptr(dXt), _lib.ptr(P0.bufs["Xt"]), _lib.ptr(g_lin), _lib.ptr(P0.bufs["lin"]),
                  _lib.ptr(P0.bufs["off.z"]), _lib.ptr(P0.bufs["emo.z"]), _lib.ptr(P0.bufs["rec_ray"]),
                  _lib.ptr(P0.bufs["rec_w"]), Ton, T, _lib.ptr(P0.buf("off.dz", 4)), _lib.ptr(P0.buf("emo.dz", 4)), s)
        src = [(self._net_bwd(P0, "off", KIND_RADIANCE, 0, 0, T, P0.bufs["off.dz"], grads["off_w"], grads["off_b"]),
                grads["off"], grads["off"], 0, T),
               (self._net_bwd(P0, "emo", KIND_RADIANCE, 88, 0, Ton, P0.bufs["emo.dz"], grads["emo_w"], grads["emo_b"]),
                grads["emo"], grads["emo"], 0, Ton)]
        self._flush_wgrad(3)
        inv, pt1 = ctx.t["inv"], ctx.t["pt1"]
        dzb, dze = self._act_batch("act_bwd", [
            dict(P=P0, z="brdf.z", out="brdf.dz", rows=8, n_ch=5, act=ACT_SIGMOID, src=g.get("etc/brdf"), inv=inv, pt1=pt1,
                 ex=[(d["d_base"], 0), (d["d_rough"], 3), (d["d_metal"], 4)]),
            dict(P=P0, z="emit.z", out="emit.dz", rows=4, n_ch=3, act=ACT_SOFTPLUS, bwd_g=d_emit_t, src=g.get("etc/emit"),
                 inv=inv, pt1=pt1, ex=[(d["d_emission"], 0)])])
        src.append((self._net_bwd(P0, "brdf", KIND_BRDF, 96, 0, T, dzb, grads["brdf_w"], grads["brdf_b"]),
                    grads["brdf"], grads["brdf"], 0, T))
        src.append((self._net_bwd(P0, "emit", KIND_EMIT, 88, 0, T, dze, grads["emit_w"], grads["emit_b"]),
                    grads["emo"], grads["emo"], 0, T))
        self._flush_wgrad(2)
        # exact normals (linear in the grid): the gradient of etc/normal is scattered inside the feature backward (same
        # samples, esr_fine_feat_bwd's grad4); etc/normal_eps sits up to several voxels away and keeps its own launch
        g4n = None
        if g.get("etc/normal") is not None:
            g4n = self._z(T * 32, 4, device=dev)
            g4n[perm, 1:4] = g["etc/normal"]

        def scatter0(src=src):
            self._march_bwd("march_bwd", P0, sp, b["rays_o"], b["rays_d"], P0.n_rays, ctx.t["off3"], dweight, g_last,
                            grads["sdf"], dsdf_extra, 1)
            self._feat_bwd(P0, ctx.scene, src, grads["sdf"], dsdf_extra=dsdf_extra, grad4=g4n)
        # (beside the perturbed heads' chain below, which does not depend on them)
        self._on_scatter_stream(2, scatter0)
        for key, noise, eps in (("etc/normal_eps", ctx.t["noise_n"], ctx.eps["normal"]),):
            if g.get(key) is None:
                continue
            g4 = self._z(T * 32, 4, device=dev)
            g4[perm, 1:4] = g[key]
            self._run("expgrad_bwd", L.esr_expgrad_bwd, sp, _lib.ptr(b["rays_o"]), _lib.ptr(b["rays_d"]),
                      _lib.ptr(P0.bufs["rec_ray"]), _lib.ptr(P0.bufs["rec_step"]), None, _lib.ptr(noise), C.c_float(eps),
                      _lib.ptr(g4), T * 32, 0, _lib.ptr(grads["sdf"]), s)

        # ---- perturbed material heads ("etc/emit_eps": PDRA's emission-smoothness loss, pdra.py:455-457)
        P3, m3 = self.epsp, ctx.t["m3"]
        T3 = P3.tiles_all
        src = []
        heads = [h for h in (("etc/emit_eps", "emit", KIND_EMIT, 88, 4, 3, ACT_SOFTPLUS, grads["emo"]),
                             ("etc/brdf_eps", "brdf", KIND_BRDF, 96, 8, 5, ACT_SIGMOID, grads["brdf"])) if g.get(h[0]) is not None]
        if heads and not ctx.t["eps_grads"]:
            raise RuntimeError(f"gradient of {heads[0][0]} requested but the forward ran with eps_grads=False")
        dzs = self._act_batch("act_bwd", [dict(P=P3, z=f"{nm}.z", out=f"{nm}.dz", rows=rows, n_ch=nch, act=act, tiles=T3,
                                               src=g[key]) for key, nm, _, _, rows, nch, act, _ in heads]) if heads else []
        for (key, nm, kind, crow, rows, nch, act, ggrid), dz in zip(heads, dzs):
            dX = self._net_bwd(P3, nm, kind, crow, 0, T3, dz, grads[f"{nm}_w"], grads[f"{nm}_b"])
            src.append((dX, None, ggrid, 0, T3))
        if src:
            self._flush_wgrad(5)
            # the points' SDF values came from esr_expgrad_fwd(pts_e, zero padding): their gradient goes back through the
            # same interpolant, inside the feature backward (grad4_mode 3 = zero_pad | own SDF-value gradient)
            self._feat_bwd(P3, ctx.scene, src, grads["sdf"], grad4_mode=3)
