"""Does a kernel return the same result every time while ANOTHER process uses the same GPU?  (tests/test_gpu_dp.py runs two ranks
on one card; tools/debug/lts_repeat.py found rare wrong results in lanes 48-63 of single waves of esr_expgrad_fwd there.)
Repeats (a) esr_expgrad_fwd on fixed explicit points and (b) a plain torch elementwise expression, compares every result
with the first one bit for bit and reports the lanes (index mod 64) of the differing elements.
   python tools/debug/two_process_lanes.py [launches] & python tools/debug/two_process_lanes.py [launches]; wait"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esr_nerf_amd import _lib                                             # noqa: E402
from esr_nerf_amd.config import lts_cfg                                   # noqa: E402
from esr_nerf_amd.esrnerf import ESRNeRF                                  # noqa: E402
from esr_nerf_amd.synthetic import init_slab_model, slab_scene            # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
role = sys.argv[2] if len(sys.argv) > 2 else "victim"        # "victim" | "aggressor" (keeps the card busy for ~N seconds) | "both"
sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=64, seed=2)
cfg = lts_cfg("cuda:0", num_2ndrays=16, num_ltspts=25)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc, seed=3)
L = _lib.lib()
scene = m.scene_struct()
sdf = m.sdf.device_view()
n = 1 << 14
g = torch.Generator(device="cuda").manual_seed(1)
lo, hi = torch.tensor(sc.xyz_min, device="cuda").float(), torch.tensor(sc.xyz_max, device="cuda").float()
pts = (lo + (hi - lo) * torch.rand(n, 3, device="cuda", generator=g)).contiguous()
x = torch.randn(n, 4, device="cuda", generator=g)
s = torch.cuda.current_stream().cuda_stream


def expgrad():
    out = torch.empty(n, 4, device="cuda")
    _lib.check(L.esr_expgrad_fwd(C.byref(scene), None, None, None, None, _lib.ptr(pts), None, C.c_float(0.0), _lib.ptr(sdf), n, 0,
                                 _lib.ptr(out), C.c_void_p(s)), "expgrad")
    return out


def plain():
    return torch.sin(x) * 3.0 + x * x


def load(seconds, stream=None):
    """Keep every CU busy: large elementwise kernels (few registers: they share SIMDs with anything) and fp32 GEMMs, alternating."""
    import time
    big = torch.randn(1 << 26, device="cuda")
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        while time.time() - t0 < seconds:
            for _ in range(20):
                big = torch.sin(big) * 1.0001
                a = (a @ a) * 1e-2
            (stream or torch.cuda.current_stream()).synchronize() if stream is None else None


def step_load(seconds, stream=None):
    """The light-transport step itself as the load (its split-fp16 MFMA kernels run one wave per SIMD at the power limit)."""
    import time
    from esr_nerf_amd.trainer import LtsStep
    scb = slab_scene("small", s_val=60.0, oblique=True, n_rays=4096, seed=2)
    bb = {k: v.cuda() for k, v in scb.batch.items()}
    bb["uncert_masks"] = (torch.arange(4096, device="cuda") % 3 == 0)
    m.train()
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        st = LtsStep(m, cfg.app.trainer, stage="lts")
        t0 = time.time()
        while time.time() - t0 < seconds:
            st.forward_loss_backward(bb, 60.0)


if role == "step":
    step_load(float(launches))
    print(f"[pid {os.getpid()}] step load done", flush=True)
    sys.exit(0)
if role == "aggressor":
    load(float(launches))
    print(f"[pid {os.getpid()}] aggressor done", flush=True)
    sys.exit(0)
side = None
if role == "both":          # the load on a second stream of THIS process
    import threading
    side = torch.cuda.Stream()
    th = threading.Thread(target=load, args=(25.0, side), daemon=True)
    th.start()
if role == "both_step":     # the light-transport step on a second stream (and thread) of THIS process
    import threading
    side = torch.cuda.Stream()
    th = threading.Thread(target=step_load, args=(25.0, side), daemon=True)
    th.start()
for name, fn in (("esr_expgrad_fwd", expgrad), ("torch elementwise", plain)):
    first = fn().clone()
    torch.cuda.synchronize()
    bad, lanes = 0, {}
    for it in range(launches):
        o = fn()
        d = (o != first).any(1)
        if bool(d.any()):
            bad += 1
            rows = d.nonzero().flatten()
            for r in rows.tolist():
                lanes[r % 64 // 16] = lanes.get(r % 64 // 16, 0) + 1
            if bad <= 3:
                r0 = int(rows[0])
                print(f"[pid {os.getpid()}] {name} launch {it}: {rows.numel()} rows differ, first {rows[:4].tolist()} now {o[r0].tolist()} first {first[r0].tolist()}", flush=True)
    print(f"[pid {os.getpid()}] {name}: {bad} of {launches} launches differ; differing rows by lane quarter (0: lanes 0-15 ... 3: lanes 48-63) {lanes}", flush=True)
