"""Host driver of the fused fine-stage path: enqueues the HIP kernels of
libesr_hip.so (include/esr_hip.h, section B) for one training step of
``VoxurfF.forward_training`` (reference: app/fine/model/voxurff.py:177-278) and
its backward.

PyTorch is plumbing here: it owns the device memory (a grow-only workspace),
the stream, and the autograd edge to the parameters.  Every arithmetic step on
the path is a HIP kernel behind the C ABI; there is no torch fallback -- if the
library is missing, `_lib.lib()` raises.

One host<->device sync per forward (the 32-byte plan header that sizes the
activation workspace); the reference has >= 8 (item() in the sampler, one per
boolean-mask compaction).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib

KIND_RADIANCE, KIND_TONEMAP = 0, 1
HID = 192
X_ROWS, DX_ROWS, XT_ROWS = 104, 64, 48


def make_scene(xyz_min, xyz_max, mask_min, mask_max, world_size, mask_size, near, stepdist,
               voxel_size, act_shift, mask_thres, fast_thres, s_val, grad_feat) -> _lib.EsrScene:
    sc = _lib.EsrScene()
    for i in range(3):
        sc.xyz_min[i], sc.xyz_max[i] = float(xyz_min[i]), float(xyz_max[i])
        sc.mask_min[i], sc.mask_max[i] = float(mask_min[i]), float(mask_max[i])
    sc.gx, sc.gy, sc.gz = [int(v) for v in world_size]
    sc.mx, sc.my, sc.mz = [int(v) for v in mask_size]
    sc.near_, sc.stepdist, sc.voxel_size = float(near), float(stepdist), float(voxel_size)
    sc.act_shift, sc.mask_thres, sc.fast_thres = float(act_shift), float(mask_thres), float(fast_thres)
    sc.s_val = float(s_val)
    # longest chord of the box in steps (+ slack for the ceil and the >=1 rule)
    diag = math.sqrt(sum((float(xyz_max[i]) - float(xyz_min[i])) ** 2 for i in range(3)))
    sc.max_steps = int(diag / float(stepdist)) + 4
    if len(grad_feat) != 4:
        raise NotImplementedError("the HIP feature kernel is built for 4 stencil radii (cfg grad_feat)")
    for i in range(4):
        sc.grad_feat[i] = float(grad_feat[i])
    return sc


_RANGE_FLAGS = {}          # device -> the split kernels' sticky range flag (FineEngine.__init__)


@dataclass
class FineCtx:
    """What the backward needs from the forward of one step."""
    scene: _lib.EsrScene
    n_rays: int
    tiles_on: int
    tiles_all: int
    counts: Dict[str, int]
    rays_o: torch.Tensor
    rays_d: torch.Tensor
    viewdirs: torch.Tensor
    off3: torch.Tensor
    mask_density: torch.Tensor
    sdf: torch.Tensor
    feat_args: object = None
    march_cache: object = None        # (ray_stats, alphainv_last, cache) of the count pass, for the backward
    x16: bool = False                 # the features of this step are the bf16 tile (ws["X16"]), not the fp32 one
    f32_only: bool = False            # the forward ran on the f32 MFMA kernels (range fallback): so must the backward


class _Workspace:
    """Grow-only device buffers, tile-major [tiles, rows, 32] fp32."""

    def __init__(self, device):
        self.device = device
        self.cap_tiles = 0
        self.buf: Dict[str, torch.Tensor] = {}

    ROWS = dict(X=X_ROWS, gnorm=4, H0=HID, H1=HID, H2=HID, z_off=4, z_emo=4, lin=4, Xt=XT_ROWS, Ht=HID,
                zt=4, rgb=4, dzt=4, dZt=HID, dXt=DX_ROWS, dz=4, dZ0=HID, dZ1=HID, dZ2=HID, dX=DX_ROWS,
                dweight=1, rec_w=1, rec_sdf=1, dsdf=1)

    def ensure(self, tiles: int):
        if tiles <= self.cap_tiles:
            return
        cap = int(max(tiles, self.cap_tiles) * 1.25) + 64          # (headroom from the first allocation on: lts_engine.Pass.ensure)
        self.buf = {k: torch.empty(cap * r * 32, dtype=torch.float32, device=self.device)
                    for k, r in self.ROWS.items()}
        for k in ("M0", "M1", "M2", "Mt"):          # ReLU sign bits, [tiles, 3, 64] u32
            self.buf[k] = torch.empty(cap * 3 * 64, dtype=torch.int32, device=self.device)
        self.buf["rec_ray"] = torch.empty(cap * 32, dtype=torch.int32, device=self.device)
        self.buf["rec_step"] = torch.empty(cap * 32, dtype=torch.int32, device=self.device)
        self.buf["X16"] = torch.empty(cap * 26 * 256, dtype=torch.uint8, device=self.device)      # bf16 input tile (esr_fine_feat_fwd_x16)
        self.cap_tiles = cap

    def __getitem__(self, k):
        return self.buf[k]


# (module-level classes: a class defined inside the method that returns it is a new TYPE per call -- a type is a reference
#  cycle of its own, its methods' closures held the engine, and the engine's workspaces then lived until Python's cycle
#  collector ran: tools/debug/cycle_probe.py)
class _F32Only:
    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        e = self.eng
        self.keep = (e.split_fwd, e.split_bwd, e.split_wgrad, e.split_tone_wgrad)
        e.split_fwd = e.split_bwd = e.split_wgrad = e.split_tone_wgrad = False

    def __exit__(self, *exc):
        e = self.eng
        e.split_fwd, e.split_bwd, e.split_wgrad, e.split_tone_wgrad = self.keep
        return False


class _PackGroup:
    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        self.eng._pack_pending = []

    def __exit__(self, et, ev, tb):
        eng = self.eng
        jobs, eng._pack_pending = eng._pack_pending, None
        if et is None and jobs:
            eng._pack_flush(jobs)
        return False


class FineEngine:
    def __init__(self, device, mlp_dtype: str = "f32"):
        """``mlp_dtype``: "f32" (f32 matrix cores; BASELINE configs C2, C4) or "bf16" (bf16 MFMA operands with fp32
        accumulation for the MLPs only -- the build-side precision choice of C3 / C5; everything else stays fp32)."""
        if mlp_dtype not in ("f32", "bf16"):
            raise ValueError("mlp_dtype must be 'f32' or 'bf16'")
        self.device = torch.device(device)
        self.bf16 = mlp_dtype == "bf16"
        self.packed16: Dict[str, torch.Tensor] = {}
        self._p16: Dict[int, C.c_void_p] = {}
        self.L = _lib.lib()
        self.ws = _Workspace(self.device)
        self.plan_dev = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.plan_host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.packed = {
            k: torch.empty(self.L.esr_mlp_packed_floats(kind), dtype=torch.float32, device=self.device)
            for k, kind in (("off", KIND_RADIANCE), ("emo", KIND_RADIANCE), ("tone", KIND_TONEMAP))}
        self.ray_bufs: Dict[int, Dict[str, torch.Tensor]] = {}
        self.wgrad_scratch = torch.empty(self.L.esr_mlp_wgrad_scratch_floats(), dtype=torch.float32,
                                         device=self.device)
        self._timing = False
        self._only = None
        self._events = []
        self.n_calls = 0
        self.overlap_wgrad = True         # the weight gradients on a second stream beside the grid scatters (bench.py turns it off
        #                                   while it times kernels one at a time)
        self._side = None
        self._raw: Dict[str, tuple] = {}
        self._pack_cache: Dict[str, tuple] = {}
        self._pack_pending = None         # inside `with self.packing():` the jobs of one esr_mlp_pack_batch launch
        self._pack_batch_cache = None
        # bf16 engine: the features are written as a bf16 tile in the operand layout of the first layer (esr_fine_feat_fwd_x16)
        # whenever the stencil radii allow it (forward())
        self.x16 = self.bf16
        # f32 engine: every MLP product on the 16-bit matrix cores from split fp16 planes, fp32 results (csrc/mlp_split.hip):
        # forward, input gradients (per-tile power-of-two scaling) and weight gradients of every net kind.  ESR_SPLIT_FWD=0:
        # the f32 MFMA kernels instead (A/B timing).  The planes' first halves are fp16: an input, a hidden activation or a
        # weight (x 64) beyond fp16's range cannot be carried.  Every split launch of a FORWARD raises a sticky device flag then
        # (the backward cannot overflow by construction: csrc/mlp.hip: split_gain_kernel), and the step that set it is RE-RUN
        # on the f32 MFMA kernels before anything leaves it (`range_probe` / `range_hit` / `f32_only`; trainer.py,
        # voxurff.py); ESR_SPLIT_STRICT=1 raises instead.
        self.split_fwd = (not self.bf16) and os.environ.get("ESR_SPLIT_FWD", "1") != "0"
        self.split_bwd = self.split_wgrad = self.split_tone_wgrad = self.split_fwd
        self.split_strict = os.environ.get("ESR_SPLIT_STRICT", "0") == "1"
        self.split_fallback_steps = 0     # steps (or image chunks) that were re-run on the f32 MFMA kernels
        self.packed_split: Dict[str, torch.Tensor] = {}
        # net kinds on the split kernels: radiance (0), tone mapper (1), BRDF (2), emission (3)
        self.split_kinds = {0, 1, 2, 3}
        self.split_kinds_bwd = set(self.split_kinds)
        self._psplit = {}
        self.range_flag = None
        self._range_host = None
        self._range_event = None
        if self.split_fwd:
            key = str(self.device)
            if key not in _RANGE_FLAGS:       # one flag per device (never freed: the library keeps its address)
                _RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.range_flag = _RANGE_FLAGS[key]
            self._range_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            with torch.cuda.device(self.device):
                _lib.check(self.L.esr_mlp_split_range_flag(_lib.ptr(self.range_flag)), "esr_mlp_split_range_flag")
        self.tone_scratch = torch.empty(self.L.esr_tone_wgrad_scratch_floats(), dtype=torch.float32, device=self.device)
        self.neus_grad = False          # cfg neus_alpha: "grad" (set by the renderer)

    # -- helpers ---------------------------------------------------------------
    def _s(self):
        return _lib.stream_ptr(self.device)

    # A ray that exceeds scene.max_steps (the march kernel's LDS bound) is skipped by the kernels and reported in the
    # plan header.  Single process: raise at once.  Data parallel (``defer_overflow``, set by the trainer steps): raising
    # on one rank would leave the others waiting in the gradient exchange, so the flag is remembered, summed over the
    # ranks with the loss, and every rank raises together (trainer._check_overflow).
    defer_overflow = False
    overflow_seen = False

    def _overflow(self):
        if not self.defer_overflow:
            raise RuntimeError("a ray exceeded scene.max_steps; the LDS bound of the march kernel is wrong")
        self.overflow_seen = True

    # -- the split-fp16 kernels' range flag: same-step fallback to the f32 MFMA kernels -------------------------------------
    _RANGE_MSG = ("an input, a hidden activation or a weight of an MLP left fp16's range (|x| >= 60000, |w| >= 1023, inf / NaN) in a "
                  "split-fp16 kernel (csrc/mlp_split.hip)")

    def range_probe(self):
        """Enqueue, on the current stream, the flag's copy to pinned host memory + an event: call it behind the LAST split
        launch of a forward.  No-op for the bf16 / f32-MFMA engines."""
        if self.range_flag is None or not self.split_fwd:
            self._range_event = None
            return
        self._range_host.copy_(self.range_flag, non_blocking=True)
        self._range_event = torch.cuda.Event()
        self._range_event.record()

    def range_hit(self) -> bool:
        """Wait for the probe (the host normally arrives here long after the device has passed it: trainer.py calls this
        with the input-gradient chain and the grid scatters queued behind the forward) and tell whether a split launch of
        this step raised the flag.  The flag is cleared; the caller re-runs the step inside ``with eng.f32_only():``."""
        ev, self._range_event = self._range_event, None
        if ev is None:
            return False
        ev.synchronize()
        if int(self._range_host[0]) == 0:
            self._range_streak = 0
            return False
        self.range_flag.zero_()
        if self.split_strict:
            if self.defer_overflow:
                # data parallel: raising here would leave the other ranks waiting in the gradient exchange.  The hit travels
                # with the march-overflow word (summed over the ranks with the loss) and EVERY rank raises at its next check
                # (trainer._check_overflow / step.close()); this step's gradients must not be used.
                self.overflow_seen = True
                self.range_strict_seen = True
                return False
            raise RuntimeError(self._RANGE_MSG + " (ESR_SPLIT_STRICT=1: no fallback)")
        import warnings
        if self.split_fallback_steps == 0:
            warnings.warn(self._RANGE_MSG + ": the step is re-run on the f32 MFMA kernels (slower; counted in "
                          "engine.split_fallback_steps)", RuntimeWarning, stacklevel=3)
        self.split_fallback_steps += 1
        self._range_streak = getattr(self, "_range_streak", 0) + 1
        if self._range_streak >= self.RANGE_STREAK_MAX and self.split_fwd:
            # the cause persists (a weight beyond 1023, a gain bound beyond 2^18): every further step would run the split attempt,
            # wait on the host and then run again on the f32 MFMA kernels -- more than twice the work.  From here on the engine
            # IS the f32-MFMA engine (what ESR_SPLIT_FWD=0 selects), and says so once.
            self.split_fwd = self.split_bwd = self.split_wgrad = self.split_tone_wgrad = False
            warnings.warn(f"{self.RANGE_STREAK_MAX} consecutive steps left fp16's range: this engine now runs every MLP launch on the "
                          "f32 MFMA kernels (as with ESR_SPLIT_FWD=0)", RuntimeWarning, stacklevel=3)
        return True

    RANGE_STREAK_MAX = 3

    def f32_only(self):
        """Context manager: every MLP launch inside runs on the f32 MFMA kernels (no range limit), on the same buffers."""
        return _F32Only(self)

    def _run(self, name, fn, *args):
        """Enqueue one C-ABI call; with timing on, bracket it with HIP events recorded on the
        stream the kernel is launched on (torch's current stream == the `stream` argument)."""
        self.n_calls += 1                     # C-ABI launches enqueued by this engine (bench.py: per step)
        if self._timing and (self._only is None or name in self._only):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self._events.append((name, e0, e1))
        else:
            rc = fn(*args)
        _lib.check(rc, name)

    def enable_timing(self, on: bool, only=None):
        """``only``: restrict the event pairs to these call names.  Bracketing all ~30 calls of a
        step costs ~2 ms of a 6.5 ms step on MI355X (measured), so the timed region of bench.py
        brackets the dominant kernel only."""
        self._timing = bool(on)
        self._only = set(only) if only is not None else None
        self._events = []

    def timing_summary(self):
        """name -> (launches, total milliseconds) over everything recorded since enable_timing(True)."""
        torch.cuda.synchronize(self.device)
        out = {}
        for name, e0, e1 in self._events:
            n, ms = out.get(name, (0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1))
        return out

    def _march(self, name, which, sp, rays_o, rays_d, viewdirs, *rest):
        """One of the three march entry points; with ``neus_grad`` (cfg neus_alpha: "grad") the variants that take the
        batch's view directions."""
        if self.neus_grad:
            fn = getattr(self.L, f"esr_fine_march_{which}_ga")
            self._run(name, fn, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(viewdirs), *rest)
        else:
            self._run(name, getattr(self.L, f"esr_fine_march_{which}"), sp, _lib.ptr(rays_o), _lib.ptr(rays_d), *rest)

    def _ray_buf(self, n, scene=None):
        if n not in self.ray_bufs:
            self.ray_bufs[n] = dict(
                cnt3=torch.empty(n, dtype=torch.int32, device=self.device),
                off3=torch.empty(n, dtype=torch.int32, device=self.device),
                stats=torch.empty(n * 3, dtype=torch.int32, device=self.device))
        rb = self.ray_bufs[n]
        if scene is not None:       # march cache (count -> fill -> backward share one walk): sized by the scene's step bound
            need = int(self.L.esr_fine_march_cache_floats(C.byref(scene), n))
            if rb.get("cache") is None or rb["cache"].numel() < need:
                rb["cache"] = torch.empty(need, dtype=torch.float32, device=self.device)
        return rb

    def pack(self, which: str, kind: int, weights: List[torch.Tensor], biases: List[torch.Tensor]):
        """Rewrite one net's reference-layout tensors into MFMA operand order.  Inside ``with eng.packing():`` the nets of
        a step are collected and go out as ONE launch (esr_mlp_pack_batch: fp32 buffers and, for the bf16 engine, their
        bf16 twins); outside, one launch per call."""
        self._raw[which] = (list(weights), list(biases))      # reference-layout tensors (esr_tone_wgrad_recompute reads them)
        # the argument struct is rebuilt (and the tensors re-validated) only when a parameter tensor moved or changed its
        # type / layout: five nets x eight tensors of marshalling per step sat on the host's critical path right before
        # the plan read.  The key holds everything the kernel assumes about a tensor (address, dtype, size, contiguity):
        # the caching allocator re-uses addresses, so an address alone does not identify a tensor.
        key = tuple((t.data_ptr(), t.dtype, t.numel(), t.is_contiguous()) for t in weights) + \
            tuple((t.data_ptr(), t.dtype, t.numel(), t.is_contiguous()) for t in biases)
        hit = self._pack_cache.get(which)
        if hit is None or hit[0] != key or hit[1] != kind or hit[6] != self.packed[which].data_ptr():
            w = _lib.EsrMlpWeights()
            for i, (a, b) in enumerate(zip(weights, biases)):
                if not (a.is_cuda and b.is_cuda and a.is_contiguous() and b.is_contiguous()
                        and a.dtype == torch.float32 and b.dtype == torch.float32):
                    raise RuntimeError("MLP parameters must be contiguous fp32 device tensors")
                w.w[i], w.b[i] = a.data_ptr(), b.data_ptr()
            p32 = _lib.ptr(self.packed[which])
            p16 = None
            if self.bf16:
                n16 = self.L.esr_mlp_packed_bf16_elems(kind)
                if which not in self.packed16 or self.packed16[which].numel() != n16:
                    self.packed16[which] = torch.empty(n16, dtype=torch.bfloat16, device=self.device)
                p16 = _lib.ptr(self.packed16[which])
                self._p16[self.packed[which].data_ptr()] = p16
            if self.split_fwd and kind in self.split_kinds:     # split fp16 planes of the weights (mlp_split.hip)
                ns = self.L.esr_mlp_packed_split_elems(kind)
                if which not in self.packed_split or self.packed_split[which].numel() != ns:
                    self.packed_split[which] = torch.empty(ns, dtype=torch.float16, device=self.device)
                self._psplit[self.packed[which].data_ptr()] = _lib.ptr(self.packed_split[which])
            hit = self._pack_cache[which] = (key, kind, w, C.byref(w), p32, p16, self.packed[which].data_ptr())
        _, _, w, wref, p32, p16, _ = hit
        if self._pack_pending is not None:
            self._pack_pending.append((which, kind, w, p32, p16))
            return
        if self.split_fwd and kind in self.split_kinds:         # (outside a packing() group: the batch entry with one job)
            return self._pack_flush([(which, kind, w, p32, p16)])
        s = self._s()
        self._run(f"mlp_pack({which})", self.L.esr_mlp_pack, kind, wref, p32, s)
        if self.bf16:
            self._run(f"mlp_pack16({which})", self.L.esr_mlp_pack_bf16, kind, wref, p16, s)

    def packing(self):
        """Context manager: the ``pack`` calls inside go out as one launch at exit."""
        return _PackGroup(self)

    def _pack_flush(self, jobs):
        split = [(self.packed_split[which].data_ptr() if (self.split_fwd and kind in self.split_kinds) else 0)
                 for which, kind, _, _, _ in jobs]
        sig = tuple((which, kind, C.addressof(w), p32.value, p16.value if p16 is not None else 0, sp)
                    for (which, kind, w, p32, p16), sp in zip(jobs, split))
        hit = self._pack_batch_cache
        if hit is None or hit[0] != sig:
            n = len(jobs)
            kinds = (C.c_int32 * n)(*[k for _, k, _, _, _ in jobs])
            ws = (C.c_void_p * n)(*[C.addressof(w) for _, _, w, _, _ in jobs])
            p32s = (C.c_void_p * n)(*[p.value for _, _, _, p, _ in jobs])
            p16s = (C.c_void_p * n)(*[(p.value if p is not None else 0) for _, _, _, _, p in jobs]) if self.bf16 else None
            psp = (C.c_void_p * n)(*split) if any(split) else None
            hit = self._pack_batch_cache = (sig, n, kinds, ws, p32s, p16s, psp)
        _, n, kinds, ws, p32s, p16s, psp = hit
        self._run("mlp_pack(all)", self.L.esr_mlp_pack_batch, n, kinds, ws, p32s, p16s, psp, self._s())

    # the three MLP entry points with the fp32 signatures; in bf16 mode the packed fp32 pointer selects its bf16 twin
    # f32 engine: a net whose split planes were packed (self._psplit: packed fp32 pointer -> planes) runs on the 16-bit matrix
    # cores with fp32 results (csrc/mlp_split.hip), any other on the f32 MFMA kernels
    def mlp_fwd(self, kind, packed, *rest):
        if not self.bf16:
            planes = self._psplit.get(packed.value) if self.split_fwd else None
            if planes is not None:
                return self.L.esr_mlp_fwd_split(kind, packed, planes, *rest)
            return self.L.esr_mlp_fwd(kind, packed, *rest)
        return self.L.esr_mlp_fwd_bf16(kind, packed, self._p16[packed.value], *rest)

    def mlp_dgrad(self, kind, packed, *rest):
        if not self.bf16:
            planes = self._psplit.get(packed.value) if (self.split_fwd and self.split_bwd and kind in self.split_kinds_bwd) else None
            if planes is not None:                      # (..., dX, stream) -> (..., dX, amax = NULL, stream)
                return self.L.esr_mlp_dgrad_split(kind, planes, *rest[:-1], None, rest[-1])
            return self.L.esr_mlp_dgrad(kind, packed, *rest)
        return self.L.esr_mlp_dgrad_bf16(kind, self._p16[packed.value], *rest)

    def mlp_wgrad(self, *args):
        return (self.L.esr_mlp_wgrad_bf16 if self.bf16 else self.L.esr_mlp_wgrad)(*args)

    def _H(self, names):
        return _lib.ptr_array([self.ws[n] for n in names])

    def feat_args(self, rays_o, rays_d, viewdirs, sdf, tiles_on, tiles_all, color_on, color_off, ws=None):
        """esr_feat_args_t for march-record sampling (the tensors must outlive the launches)."""
        ws = ws or self.ws
        fa = _lib.EsrFeatArgs()
        fa.rays_o, fa.rays_d, fa.viewdirs = rays_o.data_ptr(), rays_d.data_ptr(), viewdirs.data_ptr()
        fa.rec_ray, fa.rec_step, fa.rec_sdf = ws["rec_ray"].data_ptr(), ws["rec_step"].data_ptr(), ws["rec_sdf"].data_ptr()
        fa.sdf = sdf.data_ptr()
        for g in range(3):
            fa.color_on[g] = color_on[g].data_ptr() if color_on[g] is not None else None
            fa.color_off[g] = color_off[g].data_ptr() if color_off[g] is not None else None
        fa.tiles_on, fa.tiles_all = tiles_on, tiles_all
        return fa

    # -- forward -----------------------------------------------------------------
    def forward(self, scene, rays_o, rays_d, viewdirs, em_modes, mask_density, sdf, off_color, emo_color, prelude=None,
                heal: bool = True):
        """``_forward`` + the split kernels' range fallback.  ``heal=True`` (the autograd route: the results go to arbitrary torch
        code): wait for the forward's range probe and, when a split launch raised the flag, run the forward again on the f32
        MFMA kernels.  ``heal=False`` (trainer.FineStep): the caller examines the probe itself (``range_hit``) where the wait
        is free -- with the backward's input-gradient chain queued -- and re-runs its whole step."""
        out = self._forward(scene, rays_o, rays_d, viewdirs, em_modes, mask_density, sdf, off_color, emo_color, prelude)
        if heal and self.range_hit():
            with self.f32_only():
                out = self._forward(scene, rays_o, rays_d, viewdirs, em_modes, mask_density, sdf, off_color, emo_color, prelude)
        return out

    def _forward(self, scene, rays_o, rays_d, viewdirs, em_modes, mask_density, sdf, off_color, emo_color, prelude=None):
        """-> (ctx, alphainv_last [N], srgb_marched [N,3], lin_marched [N,3]).
        sdf [X,Y,Z], off_color/emo_color [X,Y,Z,6], mask_density [mx,my,mz]: contiguous fp32.
        ``prelude()``: enqueues work that does not depend on the march (weight packing, zeroing the gradient buffer).
        It is called between the plan kernel and the host's wait for the plan header, on a SIDE stream: the device has
        work while the host reads the survivor counts and enqueues the rest of the step, and these bandwidth-bound
        kernels then run beside the march / feature kernels instead of in front of them (tools/trace_step.py)."""
        L, s, ws = self.L, self._s(), self.ws
        n = rays_o.shape[0]
        for t in (rays_o, rays_d, viewdirs):
            if t.dtype != torch.float32:
                raise RuntimeError("rays must be fp32")
        if em_modes.dtype != torch.int64:
            raise RuntimeError("em_modes must be int64")
        cached = not self.neus_grad            # one walk per step (esr_fine_march_*_cached); not with neus_alpha "grad"
        rb = self._ray_buf(n, scene if cached else None)
        last = torch.empty(n, dtype=torch.float32, device=self.device)
        # everything of the step that starts from zero in ONE fill (each small fill is a ~5 us launch on the step's
        # critical path): plan header (8 x i32; what esr_fine_plan_begin does) | srgb | lin | the loss accumulator
        # every carve starts on a 16-byte boundary (ragged n: a future float4 access must not straddle)
        n3 = (3 * n + 3) // 4 * 4
        zb = torch.zeros(8 + 2 * n3 + 4, dtype=torch.float32, device=self.device)     # (+ the overflow flag's slot: loss_pair)
        plan_dev = zb[:8].view(torch.int32)       # this step's header; self.plan_dev stays the persistent one (plan_begin paths)
        srgb, lin = zb[8: 8 + 3 * n].view(n, 3), zb[8 + n3: 8 + n3 + 3 * n].view(n, 3)
        self._loss_acc = zb[8 + 2 * n3: 8 + 2 * n3 + 2]
        self._amax = zb[8 + 2 * n3 + 3: 8 + 2 * n3 + 4]        # max |dz| of the step (split-fp16 weight gradients)
        self._amax_t = zb[8 + 2 * n3 + 2: 8 + 2 * n3 + 3]      # max |dzt| (the tone mapper's)
        sp = C.byref(scene)
        main = torch.cuda.current_stream(self.device)
        if cached:
            self._run("march_count", L.esr_fine_march_count_cached, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(mask_density),
                      _lib.ptr(sdf), n, _lib.ptr(rb["cnt3"]), _lib.ptr(last), _lib.ptr(rb["stats"]), _lib.ptr(plan_dev),
                      _lib.ptr(rb["cache"]), s)
        else:
            self._march("march_count", "count", sp, rays_o, rays_d, viewdirs, _lib.ptr(mask_density),
                        _lib.ptr(sdf), n, _lib.ptr(rb["cnt3"]), _lib.ptr(last), _lib.ptr(rb["stats"]), _lib.ptr(plan_dev), s)
        # the counts the host waits for first (a many-workgroup sum), their copy, THEN the one-workgroup scan of the offsets,
        # which runs while the host reads the header and enqueues
        self._run("plan_totals", L.esr_fine_plan_totals, _lib.ptr(rb["cnt3"]), _lib.ptr(em_modes), _lib.ptr(rb["stats"]), n,
                  _lib.ptr(plan_dev), s)
        self.plan_host.copy_(plan_dev, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record()
        self._run("plan", L.esr_fine_plan_offsets, _lib.ptr(rb["cnt3"]), _lib.ptr(em_modes), n, _lib.ptr(rb["off3"]), _lib.ptr(plan_dev), s)
        e_pre = None
        if prelude is not None:
            # on a side stream: the packing / zeroing kernels are bandwidth-bound and run beside the march and feature
            # kernels (gather latency) instead of in front of them; the main stream joins before the first MLP launch
            side = self._side_stream(0)
            side.wait_event(landed)                                 # (also orders it behind the previous step's readers)
            with torch.cuda.stream(side):
                prelude()
                e_pre = torch.cuda.Event()
                e_pre.record(side)
        # padding lanes of the record arrays carry ray -1: filled BEFORE the wait (whole buffer of the previous step's size),
        # one dispatch less between the read-back and the first kernel that depends on it
        pre_rec = ws.buf.get("rec_ray")
        if pre_rec is not None:
            pre_rec.fill_(-1)
        landed.synchronize()                                        # the one host wait of the step
        n_on, n_off, _, _, m0, m1, m2, overflow = [int(v) for v in self.plan_host.tolist()]
        tiles_on = (n_on + 31) // 32                                # (esr_fine_plan_totals leaves the tile counts to the host)
        tiles_all = tiles_on + (n_off + 31) // 32
        if overflow & 1:                                            # (bit 1: the split kernels' range flag, informational)
            self._overflow()
        self._range_event = None
        ctx = FineCtx(scene=scene, n_rays=n, tiles_on=tiles_on, tiles_all=tiles_all,
                      counts=dict(m0=m0, m1=m1, m2=m2, m3=n_on + n_off, n_on=n_on, n_off=n_off),
                      rays_o=rays_o, rays_d=rays_d, viewdirs=viewdirs, off3=rb["off3"], mask_density=mask_density, sdf=sdf,
                      f32_only=not self.bf16 and not self.split_fwd)
        if tiles_all == 0:
            if e_pre is not None:
                main.wait_event(e_pre)
            self.range_probe()                                      # (the weight packing may have raised the flag)
            return ctx, last, srgb, lin
        ws.ensure(tiles_all)
        if ws["rec_ray"] is not pre_rec:                            # (the workspace grew: a new, unfilled buffer)
            ws["rec_ray"][: tiles_all * 32].fill_(-1)
        if cached:
            self._run("march_fill", L.esr_fine_march_fill_cached, sp, _lib.ptr(rays_o), _lib.ptr(rays_d), n, _lib.ptr(rb["off3"]),
                      _lib.ptr(rb["stats"]), _lib.ptr(rb["cache"]), _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_step"]),
                      _lib.ptr(ws["rec_w"]), _lib.ptr(ws["rec_sdf"]), s)
            ctx.march_cache = (rb["stats"], last, rb["cache"])
        else:
            self._march("march_fill", "fill", sp, rays_o, rays_d, viewdirs, _lib.ptr(mask_density),
                        _lib.ptr(sdf), n, _lib.ptr(rb["off3"]), _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_step"]),
                        _lib.ptr(ws["rec_w"]), _lib.ptr(ws["rec_sdf"]), s)
        fa = self.feat_args(rays_o, rays_d, viewdirs, sdf, tiles_on, tiles_all,
                            color_on=(emo_color, off_color, None), color_off=(off_color, None, None))
        x16 = self.x16 and all(0.0 <= float(r) <= 2.0 for r in scene.grad_feat)
        ctx.x16 = x16
        ctx.amax, ctx.amax_t = self._amax, self._amax_t             # the split weight gradients' scale sources (backward)
        if x16:
            self._run("feat_fwd", L.esr_fine_feat_fwd_x16, sp, C.byref(fa), _lib.ptr(ws["X"]), _lib.ptr(ws["gnorm"]),
                      _lib.ptr(ws["X16"]), s)
        else:
            self._run("feat_fwd", L.esr_fine_feat_fwd, sp, C.byref(fa), _lib.ptr(ws["X"]), _lib.ptr(ws["gnorm"]), s)
        ctx.feat_args = fa
        H, M = self._H(["H0", "H1", "H2"]), self._H(["M0", "M1", "M2"])
        if e_pre is not None:
            main.wait_event(e_pre)
        # the step's three radiance passes as ONE launch: the off net detached on the on-tiles (alt colour rows, nothing saved)
        # and saved on the off-tiles, the emo net on the on-tiles
        if not self.bf16 and self.split_fwd and "off" in self.packed_split and "emo" in self.packed_split:
            # products from split fp16 planes on the 16-bit matrix cores (fp32 results)
            self._run("mlp_fwd(rad)", L.esr_mlp_fwd_fine_split, _lib.ptr(self.packed["off"]), _lib.ptr(self.packed_split["off"]),
                      _lib.ptr(self.packed["emo"]), _lib.ptr(self.packed_split["emo"]), _lib.ptr(ws["X"]), tiles_on, tiles_all,
                      H, M, 88, _lib.ptr(ws["z_off"]), _lib.ptr(ws["z_emo"]), s)
        elif not self.bf16:
            self._run("mlp_fwd(rad)", L.esr_mlp_fwd_fine, _lib.ptr(self.packed["off"]), _lib.ptr(self.packed["emo"]),
                      _lib.ptr(ws["X"]), tiles_on, tiles_all, H, M, 88, _lib.ptr(ws["z_off"]), _lib.ptr(ws["z_emo"]), s)
        else:                       # bf16 engine (a workgroup = one pass's weights in LDS)
            po, pe = _lib.ptr(self.packed["off"]), _lib.ptr(self.packed["emo"])
            self._run("mlp_fwd(rad)", L.esr_mlp_fwd_fine_bf16, po, self._p16[po.value], pe, self._p16[pe.value], _lib.ptr(ws["X"]),
                      _lib.ptr(ws["X16"]) if x16 else None, tiles_on, tiles_all, H, M, 88, _lib.ptr(ws["z_off"]),
                      _lib.ptr(ws["z_emo"]), s)
        self._run("tone_in_fwd", L.esr_fine_tone_in_fwd, _lib.ptr(ws["z_off"]), _lib.ptr(ws["z_emo"]), tiles_on, tiles_all,
                                          _lib.ptr(ws["lin"]), _lib.ptr(ws["Xt"]), s)
        self._run("mlp_fwd(tone)", self.mlp_fwd, KIND_TONEMAP, _lib.ptr(self.packed["tone"]), _lib.ptr(ws["Xt"]), 0, tiles_all,
                                 self._H(["Ht"]), self._H(["Mt"]), 2, 0, _lib.ptr(ws["zt"]), s)      # (masks only: tone_wgrad.hip recomputes Ht)
        self.range_probe()                                          # behind the forward's last split launch
        self._run("composite_fwd", L.esr_fine_composite_fwd, _lib.ptr(ws["zt"]), _lib.ptr(ws["lin"]), _lib.ptr(ws["rec_ray"]),
                                            _lib.ptr(ws["rec_w"]), tiles_all, _lib.ptr(ws["rgb"]),
                                            _lib.ptr(srgb), _lib.ptr(lin), s)
        return ctx, last, srgb, lin

    # -- image rendering ---------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, *args):
        """``_evaluate`` + the split kernels' range fallback (one more run on the f32 MFMA kernels when the flag was raised)."""
        out = self._evaluate(*args)
        if self.range_hit():
            with self.f32_only():
                out = self._evaluate(*args)
        return out

    def _evaluate(self, scene, rays_o, rays_d, viewdirs, mask_density, sdf, off_color, emo_color, pos_rt, far,
                  em_mode: int):
        """``VoxurfF.forward_evaluate`` (voxurff.py:280-461), forward only: the off / emo / on radiance variants
        tone-mapped separately, depth, disparity and camera-space normals.  Returns the reference's 12 keys."""
        L, s, ws, dev = self.L, self._s(), self.ws, self.device
        n = rays_o.shape[0]
        rb = self._ray_buf(n)
        last = torch.empty(n, dtype=torch.float32, device=dev)
        sp = C.byref(scene)
        em0 = torch.zeros(n, dtype=torch.int64, device=dev)            # tile partition only: every tile "off"
        self._run("plan_begin", L.esr_fine_plan_begin, _lib.ptr(self.plan_dev), s)
        self._march("march_count", "count", sp, rays_o, rays_d, viewdirs, _lib.ptr(mask_density),
                    _lib.ptr(sdf), n, _lib.ptr(rb["cnt3"]), _lib.ptr(last), _lib.ptr(rb["stats"]), _lib.ptr(self.plan_dev), s)
        self._run("plan", L.esr_fine_plan, _lib.ptr(rb["cnt3"]), _lib.ptr(em0), _lib.ptr(rb["stats"]), n, _lib.ptr(rb["off3"]),
                  _lib.ptr(self.plan_dev), s)
        self.plan_host.copy_(self.plan_dev, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        _, _, _, T, m0, m1, m2, overflow = [int(v) for v in self.plan_host.tolist()]
        self._range_event = None
        if overflow & 1:
            raise RuntimeError("a ray exceeded scene.max_steps; the LDS bound of the march kernel is wrong")
        z3 = lambda: torch.zeros(n, 3, dtype=torch.float32, device=dev)
        out = {f"{sp_}/{v}_rgb": z3() for v in ("off", "on", "emo") for sp_ in ("srgb", "lin")}
        normal_m, depth3 = z3(), z3()
        depth = torch.zeros(n, dtype=torch.float32, device=dev)
        disp = torch.empty(n, dtype=torch.float32, device=dev)
        if T:
            ws.ensure(T)
            ws["rec_ray"][: T * 32].fill_(-1)
            self._march("march_fill", "fill", sp, rays_o, rays_d, viewdirs, _lib.ptr(mask_density),
                        _lib.ptr(sdf), n, _lib.ptr(rb["off3"]), _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_step"]),
                        _lib.ptr(ws["rec_w"]), _lib.ptr(ws["rec_sdf"]), s)
            fa = self.feat_args(rays_o, rays_d, viewdirs, sdf, 0, T, color_on=(None, None, None),
                                color_off=(off_color, emo_color, None))
            self._run("feat_fwd", L.esr_fine_feat_fwd, sp, C.byref(fa), _lib.ptr(ws["X"]), _lib.ptr(ws["gnorm"]), s)
            H, M = self._H(["H0", "H1", "H2"]), self._H(["M0", "M1", "M2"])
            for net, crow, z in (("off", 0, "z_off"), ("emo", 88, "z_emo")):
                self._run(f"mlp_fwd({net})", self.mlp_fwd, KIND_RADIANCE, _lib.ptr(self.packed[net]), _lib.ptr(ws["X"]), 0, T,
                          H, M, 0, crow, _lib.ptr(ws[z]), s)
            for name, za, zb, ton in (("off", "z_off", "z_emo", 0), ("emo", "z_emo", "z_emo", 0), ("on", "z_off", "z_emo", T)):
                self._run("tone_in_fwd", L.esr_fine_tone_in_fwd, _lib.ptr(ws[za]), _lib.ptr(ws[zb]), ton, T,
                          _lib.ptr(ws["lin"]), _lib.ptr(ws["Xt"]), s)
                self._run("mlp_fwd(tone)", self.mlp_fwd, KIND_TONEMAP, _lib.ptr(self.packed["tone"]), _lib.ptr(ws["Xt"]), 0, T,
                          self._H(["Ht"]), self._H(["Mt"]), 0, 0, _lib.ptr(ws["zt"]), s)
                self._run("composite_fwd", L.esr_fine_composite_fwd, _lib.ptr(ws["zt"]), _lib.ptr(ws["lin"]),
                          _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_w"]), T, _lib.ptr(ws["rgb"]),
                          _lib.ptr(out[f"srgb/{name}_rgb"]), _lib.ptr(out[f"lin/{name}_rgb"]), s)
            aux = torch.empty(T * 8 * 32, dtype=torch.float32, device=dev)
            rt = (C.c_float * 9)(*[float(v) for v in pos_rt.detach().cpu().reshape(-1).tolist()])
            self._run("eval_aux", L.esr_eval_aux, _lib.ptr(ws["X"]), X_ROWS, 40, 36, 32, _lib.ptr(ws["rec_ray"]),
                      _lib.ptr(ws["rec_step"]), T, rt, C.c_float(scene.stepdist), _lib.ptr(aux), s)
            self._run("composite3_fwd(normal)", L.esr_composite3_fwd, _lib.ptr(aux), 8, _lib.ptr(ws["rec_ray"]),
                      _lib.ptr(ws["rec_w"]), T, _lib.ptr(normal_m), s)
            self._run("composite3_fwd(depth)", L.esr_composite3_fwd, C.c_void_p(aux.data_ptr() + 4 * 32 * 4), 8,
                      _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_w"]), T, _lib.ptr(depth3), s)
        self._run("eval_disp", L.esr_eval_disp, _lib.ptr(depth3), _lib.ptr(last), C.c_float(far), n, _lib.ptr(depth),
                  _lib.ptr(disp), s)
        self.range_probe()
        out.update({"etc/depth": depth, "etc/disp": disp, "etc/normal": normal_m, "etc/white_bg": last.unsqueeze(-1)})
        pick = "off" if int(em_mode) == 0 else "on"
        out["srgb/rgb"], out["lin/rgb"] = out[f"srgb/{pick}_rgb"], out[f"lin/{pick}_rgb"]
        self.last_eval_counts = dict(m0=m0, m1=m1, m2=m2, tiles=T)
        return out

    # -- backward ----------------------------------------------------------------
    def backward(self, ctx: FineCtx, g_last, g_srgb, g_lin, grads: Dict[str, Optional[torch.Tensor]],
                 after_grids=None):
        """Accumulates into the (zero-initialised, reference-layout) tensors of ``grads``:
        sdf [X,Y,Z], off_color/emo_color [X,Y,Z,6], off_w/off_b/emo_w/emo_b (lists of 4),
        tone_w/tone_b (lists of 2).

        Order: input-gradient chain, then two independent branches -- the grid scatters (feat_bwd,
        march_bwd: LDS / L2 atomics) and the weight gradients (matrix cores).  With ``overlap_wgrad``
        the weight gradients run on a second HIP stream beside the scatters and are joined at the end.
        ``after_grids()`` is called once the grid gradients are complete in stream order: the
        data-parallel step starts the (large) grid all-reduce there, underneath the wgrad kernels.
        A forward that ran on the f32 MFMA kernels (the range fallback) gets an f32 backward."""
        if ctx.f32_only and self.split_fwd:
            with self.f32_only():
                return self._backward(ctx, g_last, g_srgb, g_lin, grads, after_grids)
        return self._backward(ctx, g_last, g_srgb, g_lin, grads, after_grids)

    def _backward(self, ctx: FineCtx, g_last, g_srgb, g_lin, grads, after_grids=None):
        L, ws = self.L, self.ws
        sp = C.byref(ctx.scene)
        to, ta = ctx.tiles_on, ctx.tiles_all
        g_last, g_srgb, g_lin = g_last.contiguous(), g_srgb.contiguous(), g_lin.contiguous()
        main = torch.cuda.current_stream(self.device)
        s = self._s()
        dweight = ws["dweight"] if ta > 0 else torch.zeros(32, dtype=torch.float32, device=self.device)
        # Queues (HIP streams of this device):
        #   main     composite_bwd -> march_bwd -> dgrad(tone) -> tone_in_bwd -> dgrad(rad) -> feat_bwd
        #   wgrad    (overlap_wgrad) the weight gradients, after dgrad(rad), beside the feature scatter
        # (a third stream for the scatters was measured slower on MI355X -- C2: 4.08 ms without, 4.19-4.47 ms with: the
        #  atomics-heavy scatters slow the matrix kernels more than they hide -- and is gone)
        overlap = self.overlap_wgrad and ta > 0
        split = not self.bf16 and self.split_fwd and self.split_bwd

        # the march backward's value-tap gradients of the recorded samples ride on the feature backward's SDF window
        # (ws["dsdf"]) instead of 8 L2 atomics each; not with neus_alpha "grad" (its gradient taps scatter anyway)
        fold = ta > 0 and not self.neus_grad and grads.get("sdf") is not None

        def march_bwd(s_):
            if fold and ctx.march_cache is not None:
                st, la, ca = ctx.march_cache
                self._run("march_bwd", L.esr_fine_march_bwd_cached, sp, _lib.ptr(ctx.rays_o), _lib.ptr(ctx.rays_d), ctx.n_rays,
                          _lib.ptr(ctx.off3), _lib.ptr(st), _lib.ptr(la), _lib.ptr(ca), _lib.ptr(dweight), _lib.ptr(g_last),
                          _lib.ptr(grads["sdf"]), _lib.ptr(ws["dsdf"]), 0, s_)
                return
            if fold:
                self._run("march_bwd", L.esr_fine_march_bwd_rec, sp, _lib.ptr(ctx.rays_o), _lib.ptr(ctx.rays_d),
                          _lib.ptr(ctx.mask_density), _lib.ptr(ctx.sdf), ctx.n_rays, _lib.ptr(ctx.off3), _lib.ptr(dweight),
                          _lib.ptr(g_last), _lib.ptr(grads["sdf"]), _lib.ptr(ws["dsdf"]), 0, s_)
                return
            self._march("march_bwd", "bwd", sp, ctx.rays_o, ctx.rays_d, ctx.viewdirs, _lib.ptr(ctx.mask_density),
                        _lib.ptr(ctx.sdf), ctx.n_rays, _lib.ptr(ctx.off3), _lib.ptr(dweight), _lib.ptr(g_last),
                        _lib.ptr(grads["sdf"]), s_)

        def feat_bwd(s_):
            src = (_lib.EsrFeatBwdSrc * 1)()
            src[0].dX = ws["dX"].data_ptr()
            src[0].grad_color_on = grads["emo_color"].data_ptr()      # on-tiles carry the emo net's gradient
            src[0].grad_color_off = grads["off_color"].data_ptr()
            src[0].t0, src[0].t1 = 0, ta
            self._run("feat_bwd", L.esr_fine_feat_bwd, sp, C.byref(ctx.feat_args), _lib.ptr(ws["X"]),
                      _lib.ptr(ws["gnorm"]), src, 1, _lib.ptr(ws["dsdf"]) if fold else None, _lib.ptr(grads["sdf"]),
                      None, None, 0, s_)

        def tone_wgrad(s_):
            # from Xt and dzt alone: the hidden layer is recomputed inside (tone_wgrad.hip)
            (w0, w1), (b0, _) = self._raw["tone"]
            if split and self.split_tone_wgrad:
                # products on the 16-bit matrix cores; the gradient operand's scale: ctx.amax_t, left behind by the split
                # input-gradient kernel (max |dzt| x the net's gain bound: no overflow by construction)
                self._run("tone_wgrad", L.esr_tone_wgrad_recompute_split, _lib.ptr(ws["Xt"]), _lib.ptr(ws["dzt"]), _lib.ptr(w0.detach()),
                          _lib.ptr(b0.detach()), _lib.ptr(w1.detach()), _lib.ptr(ctx.amax_t), 0, ta, _lib.ptr(grads["tone_w"][0]),
                          _lib.ptr(grads["tone_b"][0]), _lib.ptr(grads["tone_w"][1]), _lib.ptr(grads["tone_b"][1]),
                          _lib.ptr(self.tone_scratch), C.c_int64(self.tone_scratch.numel()), s_)
                return
            self._run("tone_wgrad", L.esr_tone_wgrad_recompute_bf16 if self.bf16 else L.esr_tone_wgrad_recompute,
                      _lib.ptr(ws["Xt"]), _lib.ptr(ws["dzt"]), _lib.ptr(w0.detach()),
                      _lib.ptr(b0.detach()), _lib.ptr(w1.detach()), 0, ta, _lib.ptr(grads["tone_w"][0]),
                      _lib.ptr(grads["tone_b"][0]), _lib.ptr(grads["tone_w"][1]), _lib.ptr(grads["tone_b"][1]),
                      _lib.ptr(self.tone_scratch), C.c_int64(self.tone_scratch.numel()), s_)

        def wgrads(s_):
            # the tone mapper's by recomputation, then one call for the two radiance nets: layers of the same kernel shape
            # share a launch (esr_mlp_wgrad_batch)
            tone_wgrad(s_)
            Hh, dZh = self._H(["H0", "H1", "H2"]), self._H(["dZ0", "dZ1", "dZ2"])
            keep = [Hh, dZh]
            todo = [(KIND_RADIANCE, ws["X"], Hh, dZh, ws["dz"], 0, to, "emo_w", "emo_b"),
                    (KIND_RADIANCE, ws["X"], Hh, dZh, ws["dz"], to, ta, "off_w", "off_b")]
            jobs = (_lib.EsrWgradJob * len(todo))()
            for j, (kind, X, Hs, dZs, dzs, r0, r1, gwk, gbk) in enumerate(todo):
                gwa, gba = _lib.ptr_array(grads[gwk]), _lib.ptr_array(grads[gbk])
                keep += [gwa, gba]
                jb = jobs[j]
                jb.kind, jb.color_row0, jb.t0, jb.t1 = kind, 0, r0, r1
                jb.X, jb.dz = X.data_ptr(), dzs.data_ptr()
                jb.H, jb.dZ = C.addressof(Hs), C.addressof(dZs)
                jb.gw, jb.gb = C.addressof(gwa), C.addressof(gba)
                if ctx.x16:
                    jb.X16 = ws["X16"].data_ptr()
                if split and self.split_wgrad:          # (non-NULL amax selects the split-fp16 weight-gradient kernel)
                    jb.amax = ctx.amax.data_ptr()
            self._run("mlp_wgrad(all)", L.esr_mlp_wgrad_batch, jobs, len(todo), 1 if self.bf16 else 0,
                      _lib.ptr(self.wgrad_scratch), C.c_int64(self.wgrad_scratch.numel()), s_)

        if ta == 0:
            march_bwd(s)
            if after_grids is not None:
                after_grids()
            return
        self._run("composite_bwd", L.esr_fine_composite_bwd, _lib.ptr(g_srgb), _lib.ptr(g_lin), _lib.ptr(ws["rgb"]),
                  _lib.ptr(ws["lin"]), _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_w"]), ta, _lib.ptr(ws["dweight"]),
                  _lib.ptr(ws["dzt"]), s)
        if fold:
            march_bwd(s)
        dZt_arg = _lib.ptr_array([None])            # (the tone mapper's hidden gradient is recomputed, not stored)
        M = self._H(["M0", "M1", "M2"])
        dZ = self._H(["dZ0", "dZ1", "dZ2"])
        if self.bf16:
            self._run("mlp_dgrad(tone)", L.esr_mlp_dgrad_bf16, KIND_TONEMAP, self._p16[self.packed["tone"].data_ptr()],
                      _lib.ptr(ws["dzt"]), 0, ta, self._H(["Mt"]), dZt_arg, _lib.ptr(ws["dXt"]), s)
        elif split:
            # (the split kernels leave max |dzt| / max |dz| x the nets' gain bounds behind: the weight gradients' scales)
            self._run("mlp_dgrad(tone)", L.esr_mlp_dgrad_split, KIND_TONEMAP, _lib.ptr(self.packed_split["tone"]), _lib.ptr(ws["dzt"]),
                      0, ta, self._H(["Mt"]), dZt_arg, _lib.ptr(ws["dXt"]), _lib.ptr(ctx.amax_t), s)
        else:
            self._run("mlp_dgrad(tone)", L.esr_mlp_dgrad, KIND_TONEMAP, _lib.ptr(self.packed["tone"]), _lib.ptr(ws["dzt"]), 0, ta,
                      self._H(["Mt"]), dZt_arg, _lib.ptr(ws["dXt"]), s)
        self._run("tone_in_bwd", L.esr_fine_tone_in_bwd, _lib.ptr(ws["dXt"]), _lib.ptr(ws["Xt"]), _lib.ptr(g_lin), _lib.ptr(ws["lin"]),
                  _lib.ptr(ws["z_off"]), _lib.ptr(ws["z_emo"]), _lib.ptr(ws["rec_ray"]), _lib.ptr(ws["rec_w"]), to, ta,
                  _lib.ptr(ws["dz"]), s)
        # both radiance nets' input gradients: one launch
        if self.bf16:
            pe, po = self.packed["emo"].data_ptr(), self.packed["off"].data_ptr()
            self._run("mlp_dgrad(rad)", L.esr_mlp_dgrad_fine_bf16, self._p16[pe], self._p16[po],
                      _lib.ptr(ws["dz"]), to, ta, M, dZ, _lib.ptr(ws["dX"]), s)
        elif split:
            self._run("mlp_dgrad(rad)", L.esr_mlp_dgrad_fine_split, _lib.ptr(self.packed_split["emo"]),
                      _lib.ptr(self.packed_split["off"]), _lib.ptr(ws["dz"]), to, ta, M, dZ, _lib.ptr(ws["dX"]),
                      _lib.ptr(ctx.amax), s)
        else:
            self._run("mlp_dgrad(rad)", L.esr_mlp_dgrad_fine, _lib.ptr(self.packed["emo"]), _lib.ptr(self.packed["off"]),
                      _lib.ptr(ws["dz"]), to, ta, M, dZ, _lib.ptr(ws["dX"]), s)
        if not overlap:
            feat_bwd(s)
            if not fold:
                march_bwd(s)
            if after_grids is not None:
                after_grids()
            wgrads(s)
            return
        e_dgrad = torch.cuda.Event()
        e_dgrad.record(main)
        side = self._side_stream(0)
        side.wait_event(e_dgrad)
        with torch.cuda.stream(side):
            wgrads(_lib.stream_ptr(self.device))
            e_w = torch.cuda.Event()
            e_w.record(side)
        feat_bwd(s)
        if not fold:
            march_bwd(s)
        if after_grids is not None:
            after_grids()
        main.wait_event(e_w)

    def _side_stream(self, i=0):
        if self._side is None:
            self._side = {}
        if i not in self._side:
            self._side[i] = torch.cuda.Stream(self.device)
        return self._side[i]

    # -- fused trainer-step loss (app/fine/fine.py:355-382) ------------------------
    def loss_fwd_bwd(self, last, srgb, lin, rgbs, white_bg=True, weight_linear=0.1, weight_entropy_last=0.001,
                     scale=1.0):
        """``scale``: multiplies the loss and its gradients inside the kernel (a data-parallel rank's share of the batch)."""
        n = last.shape[0]
        acc = getattr(self, "_loss_acc", None)           # zeroed with the step's other accumulators (forward)
        self._loss_acc = None
        pair = acc if acc is not None and acc.device == srgb.device else torch.zeros(2, dtype=torch.float32, device=self.device)
        # [loss, 0]: the data-parallel step all-reduces this pair as it is (second slot: its overflow flag) instead of
        # building one with a fill and a cat at the end of the step
        self.loss_pair = pair
        loss = pair[:1]
        g_srgb = torch.empty_like(srgb)
        g_lin = torch.empty_like(lin)
        g_last = torch.empty_like(last)
        self._run("loss", self.L.esr_fine_loss_fwd_bwd_dp,
                  _lib.ptr(srgb), _lib.ptr(lin), _lib.ptr(last), _lib.ptr(rgbs.contiguous()), n,
                  C.c_float(1.0 if white_bg else 0.0), C.c_float(weight_linear), C.c_float(weight_entropy_last),
                  C.c_float(scale), _lib.ptr(loss), _lib.ptr(g_srgb), _lib.ptr(g_lin), _lib.ptr(g_last), self._s())
        return loss, g_last, g_srgb, g_lin
