"""Diagnostic (GPU box): where do the full-size colour-grid gradient differences against the oracle sit?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
torch.set_num_threads(16)
from test_gpu_full_size import _fine_model, _fine_oracle, _run_fine, _oracle_fine
from esr_nerf_amd.synthetic import slab_scene
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = slab_scene(name, s_val=20.0)
m = _fine_model(sc)
loss, grads = _run_fine(m, sc, 20.0)
fp, c, P = _fine_oracle(m, sc)
res, o_loss, keep = _oracle_fine(fp, c, P, sc, 20.0)
knife = keep["knife"]
print("samples", knife.numel(), "knife<1e-5", int((knife < 1e-5).sum()), "<2e-6", int((knife < 2e-6).sum()), "<5e-7", int((knife < 5e-7).sum()))
pts = keep["pts"]
dims = torch.tensor([int(v) for v in c.world_size])
idx = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min) * (dims - 1)
i0 = idx.floor().long()
for gname in ("off_color.grid", "emo_color.grid", "sdf.grid"):
    g, o = grads[gname].cpu(), P[gname].grad
    err = (g - o).abs()
    if g.shape[1] > 1:
        err = err.amax(1, keepdim=True)
    mx = float(o.abs().max())
    bad = err[0, 0] > 1e-4 * mx
    print(gname, "max", mx, "worst rel", float(err.max()) / mx, "bad cells", int(bad.sum()), "touched", int((o.abs().amax(1)[0] > 0).sum()))
    for thr in (1e-5, 2e-6, 5e-7):
        ks = knife < thr
        mark = torch.zeros(tuple(dims.tolist()), dtype=torch.bool)
        rad = 0 if g.shape[1] > 1 else 2
        p0 = i0[ks]
        for dx in range(-rad, rad + 2):
            for dy in range(-rad, rad + 2):
                for dz in range(-rad, rad + 2):
                    q = p0 + torch.tensor([dx, dy, dz])
                    q = torch.minimum(torch.maximum(q, torch.zeros(3, dtype=torch.long)), dims - 1)
                    mark[q[:, 0], q[:, 1], q[:, 2]] = True
        print("   thr", thr, "knife samples", int(ks.sum()), "marked cells", int(mark.sum()), "bad cells NOT marked", int((bad & ~mark).sum()),
              "worst rel outside marks", float(err[0, 0][~mark].max()) / mx)

# ---- unexplained cells: which samples touch them?
g, o = grads["emo_color.grid"].cpu(), P["emo_color.grid"].grad
err = (g - o).abs().amax(1)[0] / float(o.abs().max())
ks = knife < 5e-7
mark = torch.zeros(tuple(dims.tolist()), dtype=torch.bool)
p0 = i0[ks]
for dx in range(0, 2):
    for dy in range(0, 2):
        for dz in range(0, 2):
            q = torch.minimum(p0 + torch.tensor([dx, dy, dz]), dims - 1)
            mark[q[:, 0], q[:, 1], q[:, 2]] = True
bad = (err > 1e-4) & ~mark
print("unexplained cells", int(bad.sum()))
cells = bad.nonzero()[:12]
w = keep["weights"].detach()
for cell in cells:
    d = (i0 - cell)
    near = ((d >= -1) & (d <= 0)).all(-1)
    idxs = near.nonzero()[:, 0]
    print("cell", cell.tolist(), "err", float(err[tuple(cell.tolist())]), "g", g[0, :, cell[0], cell[1], cell[2]].tolist()[:3], "o", o[0, :, cell[0], cell[1], cell[2]].tolist()[:3])
    for j in idxs[:6]:
        print("    sample", int(j), "ray", int(keep["ray_id"][j]), "step", int(keep["step_id"][j]), "knife", float(knife[j]), "w", float(w[j]))
