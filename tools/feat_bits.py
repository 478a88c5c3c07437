"""Dump the X tiles (feature rows) of one forward of a test scene: run under two builds, compare bit for bit.
  python tools/feat_bits.py OUT.npy [name oblique s_val n_rays]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from tests.test_gpu_fine_path import build_gpu_model, gpu_batch
from esr_nerf_amd.synthetic import slab_scene
out = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "small"
obl = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
s_val = float(sys.argv[4]) if len(sys.argv) > 4 else 45.0
n_rays = int(sys.argv[5]) if len(sys.argv) > 5 else 300
sc = slab_scene(name, s_val=s_val, oblique=obl, n_rays=n_rays, seed=3, mask="full")
m = build_gpu_model(sc, seed=1, grid_seed=2)
b = gpu_batch(sc)
res = m(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=s_val)
torch.cuda.synchronize()
ws = m.engine.ws
T = (m.last_counts["m3"] + 63) // 32 + 2
X = ws["X"][: T * 104 * 32].cpu().numpy().copy()
np.save(out, X)
print(out, X.shape, float(np.abs(X).sum()))
