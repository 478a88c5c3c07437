"""Experiment (round 6): would a CHUNKED backward -- input gradients of a tile range, then at once the weight gradients of that
range -- let the weight-gradient launch read dZ from the 256 MiB memory-side cache instead of HBM?
(tools/debug/mall_after_write.py: freshly written data up to 256 MB reads back ~2x faster.)
   python tools/debug/chunked_wgrad.py      (GPU box; C2's shape, the split-fp16 engine's kernels, after one real step)
Prints: the two full launches (serialised), and for K = 2..32 chunks the time of the alternating sequence."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from esr_nerf_amd import _lib                                             # noqa: E402
from esr_nerf_amd.fine_engine import KIND_RADIANCE                        # noqa: E402
from esr_nerf_amd.synthetic import slab_scene                             # noqa: E402
from esr_nerf_amd.trainer import FineStep                                 # noqa: E402
from test_gpu_fine_path import build_gpu_model, gpu_batch                 # noqa: E402

sc = slab_scene("C2", s_val=20.0)
m = build_gpu_model(sc, seed=1, grid_seed=2)
b = gpu_batch(sc)
step = FineStep(m)
for _ in range(3):
    loss, grads = step.forward_loss_backward(b, 20.0)
torch.cuda.synchronize()
eng = m.engine
L, ws = eng.L, eng.ws
lc = m.last_counts
to, ta = (lc["n_on"] + 31) // 32, (lc["n_on"] + 31) // 32 + (lc["n_off"] + 31) // 32
print("tiles on / all:", to, ta)
s = _lib.stream_ptr("cuda:0")
M, dZ, H = eng._H(["M0", "M1", "M2"]), eng._H(["dZ0", "dZ1", "dZ2"]), eng._H(["H0", "H1", "H2"])
amax = torch.zeros(1, device="cuda")
gw = {k: [torch.zeros_like(p) for p in v] for k, v in (("emo_w", grads_w) for grads_w in [[g for n, g in grads.items() if n.startswith("emo_rgbnet") and n.endswith("weight")]])}
names = lambda net, suf: [g for n, g in grads.items() if n.startswith(net) and n.endswith(suf)]
G = {"emo_w": [torch.zeros_like(g) for g in names("emo_rgbnet", "weight")], "emo_b": [torch.zeros_like(g) for g in names("emo_rgbnet", "bias")],
     "off_w": [torch.zeros_like(g) for g in names("off_rgbnet", "weight")], "off_b": [torch.zeros_like(g) for g in names("off_rgbnet", "bias")]}
keep = []


def dgrad(t0, t1):
    for net, lo, hi in (("emo", 0, to), ("off", to, ta)):
        a, e = max(t0, lo), min(t1, hi)
        if e > a:
            _lib.check(L.esr_mlp_dgrad_split(KIND_RADIANCE, _lib.ptr(eng.packed_split[net]), _lib.ptr(ws["dz"]), a, e, M, dZ, _lib.ptr(ws["dX"]),
                                             _lib.ptr(amax), s), "dgrad")


def wgrad(t0, t1):
    todo = []
    for net, lo, hi in (("emo", 0, to), ("off", to, ta)):
        a, e = max(t0, lo), min(t1, hi)
        if e > a:
            todo.append((net, a, e))
    jobs = (_lib.EsrWgradJob * len(todo))()
    for j, (net, a, e) in enumerate(todo):
        gwa, gba = _lib.ptr_array(G[net + "_w"]), _lib.ptr_array(G[net + "_b"])
        keep.extend([gwa, gba])
        jb = jobs[j]
        jb.kind, jb.color_row0, jb.t0, jb.t1 = KIND_RADIANCE, 0, a, e
        jb.X, jb.dz = ws["X"].data_ptr(), ws["dz"].data_ptr()
        jb.H, jb.dZ = C.addressof(H), C.addressof(dZ)
        jb.gw, jb.gb = C.addressof(gwa), C.addressof(gba)
        jb.amax = amax.data_ptr()
    _lib.check(L.esr_mlp_wgrad_batch(jobs, len(todo), 0, _lib.ptr(eng.wgrad_scratch), C.c_int64(eng.wgrad_scratch.numel()), s), "wgrad")


def timed(fn, reps=6):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


dgrad(0, ta); torch.cuda.synchronize()                                  # (amax)
t_d, t_w = timed(lambda: dgrad(0, ta)), timed(lambda: wgrad(0, ta))
print(f"full launches: input gradients {t_d:.3f} ms, weight gradients {t_w:.3f} ms, sum {t_d + t_w:.3f} ms")
both = timed(lambda: (dgrad(0, ta), wgrad(0, ta)))
print(f"back to back in one timed region: {both:.3f} ms")
for K in (2, 4, 8, 16, 32):
    c = (ta + K - 1) // K
    c = (c + 3) // 4 * 4

    def seq():
        for t0 in range(0, ta, c):
            dgrad(t0, min(t0 + c, ta))
            wgrad(t0, min(t0 + c, ta))
    tw_only = timed(lambda: [wgrad(t0, min(t0 + c, ta)) for t0 in range(0, ta, c)])
    td_only = timed(lambda: [dgrad(t0, min(t0 + c, ta)) for t0 in range(0, ta, c)])
    print(f"K = {K:2d} chunks of {c} tiles ({c * 72 / 1024:.0f} MB of dZ each): alternating {timed(seq):.3f} ms; the {K} input-gradient launches alone "
          f"{td_only:.3f}, the {K} weight-gradient launches alone (dZ cold) {tw_only:.3f}")
