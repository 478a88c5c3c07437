"""Time the IMPORTED reference (VoxurfF.forward_training + the Fine.learn loss arithmetic + backward) on the C2 slab
scene in the build container -- TEST INFRASTRUCTURE, build container only (cross-check of the
CPU port's speed).  The three native ops are served by the C oracle (oracle/ref_import.py).

    python -m oracle.time_reference [n_rays] [iters]
"""
import sys
import time

import numpy as np
import torch

from esr_nerf_amd.config import fine_cfg
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from oracle import fine_path as fp
from oracle import ref_import
from oracle.gen_golden import reference_loss

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ns = ref_import.load()
cfg = fine_cfg("cpu")
sc = slab_scene("C2", s_val=20.0, n_rays=n_rays)
torch.manual_seed(0)
np.random.seed(0)
m = ns.VoxurfF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
               sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc)
m.train()
b = sc.batch
c = fp.make_consts(cfg.app.model, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
                   sc.mask_density, sc.near, sc.num_voxels)
P = fp.params_from_state_dict({k: v.detach() for k, v in m.state_dict().items()})
print(f"threads {torch.get_num_threads()}, rays {n_rays}")
for name in ("reference", "port"):
    ts = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        if name == "reference":
            m.zero_grad(set_to_none=True)
            res = m(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=20.0)
            loss = reference_loss(ns, dict(res), b["rgbs"], cfg)
        else:
            for v in P.values():
                v.grad = None
            res = fp.forward_training(P, c, b, 20.0)
            loss, _ = fp.fine_loss(res, b["rgbs"])
        loss.backward()
        ts.append(time.perf_counter() - t0)
    t = sum(ts[1:]) / iters
    print(f"{name:9s}: {t:.2f} s/iter = {n_rays / t:.0f} rays/s (loss {float(loss):.6f})")
