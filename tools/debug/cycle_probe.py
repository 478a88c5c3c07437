"""Which reference cycle keeps a training experiment's engine (and its device memory) alive until the cycle collector runs?
(GPU box)  python tools/debug/cycle_probe.py [stage]"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import teacher_student as ts
stage = sys.argv[1] if len(sys.argv) > 1 else "pdra"
if stage in ("lts_autograd", "lts_eval", "fine_autograd", "fine_eval"):
    # the renderers' own routes (model(...) in training mode + backward / in evaluation mode), as the drop-in boundary runs them
    import numpy as np
    from esr_nerf_amd.config import fine_cfg, lts_cfg
    from esr_nerf_amd.esrnerf import ESRNeRF
    from esr_nerf_amd.voxurff import VoxurfF
    from esr_nerf_amd.synthetic import init_slab_model, slab_scene
    sc = slab_scene("small", s_val=60.0, oblique=True, n_rays=256, seed=2)
    b = {k: v.cuda() for k, v in sc.batch.items()}
    b["uncert_masks"] = (torch.arange(256) % 2 == 0).cuda()

    def run(*_a, **_k):
        lts = stage.startswith("lts")
        cfg = (lts_cfg("cuda:0", num_2ndrays=8, num_ltspts=16) if lts else fine_cfg("cuda:0"))
        m = (ESRNeRF if lts else VoxurfF)(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                                          sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
        init_slab_model(m, sc, seed=3)
        tr = cfg.app.trainer
        kw = dict(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"])
        if stage.endswith("autograd"):
            m.train()
            if lts:
                kw.update(uncert_masks=b["uncert_masks"], normal_eps=tr.normal_eps, emit_eps=tr.emit_eps)
            res = m(em_modes=b["em_modes"], s_val=60.0, **kw)
            sum(v.sum() for v in res.values() if v.requires_grad).backward()
        else:
            m.s_val = 60.0
            m.eval()
            ekw = dict(render_pbr=False, chunk_sz=64) if lts else {}
            m(em_modes=1, pos_rt=torch.eye(3).cuda(), **ekw, **{k: v[:64].contiguous() for k, v in kw.items()})
        torch.cuda.synchronize()
else:
    run = dict(fine=ts.fine_experiment, pdra=ts.pdra_experiment, finetune=ts.finetune_experiment)[stage]
run("f32", steps=4, seed=0)
gc.collect()
gc.disable()
run("f32", steps=4, seed=1)
print("allocated before collect: %.0f MiB" % (torch.cuda.memory_allocated() / 2**20))
gc.set_debug(gc.DEBUG_SAVEALL)
gc.collect()
garbage = list(gc.garbage)
ids = {id(o): o for o in garbage}
print("garbage objects:", len(garbage))


def name(o):
    t = type(o).__name__
    if t in ("function", "method"):
        return f"{t} {getattr(o, '__qualname__', '?')}"
    if t == "cell":
        try:
            return f"cell -> {type(o.cell_contents).__name__}"
        except ValueError:
            return "cell (empty)"
    if t == "dict":
        return "dict " + str(list(o.keys())[:6])
    if t in ("tuple", "list"):
        return f"{t}[{len(o)}] " + str([type(x).__name__ for x in o[:5]])
    return t


def cycle_from(start):
    # breadth-first over referents inside the garbage set until start is reached again
    prev, queue = {id(start): None}, [start]
    while queue:
        o = queue.pop(0)
        for r in gc.get_referents(o):
            if id(r) not in ids:
                continue
            if r is start:
                path, k = [o], id(o)
                while prev[k] is not None:
                    path.append(ids[prev[k]]); k = prev[k]
                return list(reversed(path))
            if id(r) not in prev:
                prev[id(r)] = id(o); queue.append(r)
    return None


# strongly connected components of the garbage graph (iterative Tarjan): every component with more than one object (or a
# self-reference) is a cycle; what merely hangs off one is not printed
index, low, onstack, stack, comps, counter = {}, {}, set(), [], [], [0]
for root in garbage:
    if id(root) in index:
        continue
    work = [(root, iter([r for r in gc.get_referents(root) if id(r) in ids]))]
    index[id(root)] = low[id(root)] = counter[0]; counter[0] += 1; stack.append(root); onstack.add(id(root))
    while work:
        o, it = work[-1]
        advanced = False
        for r in it:
            if id(r) not in index:
                index[id(r)] = low[id(r)] = counter[0]; counter[0] += 1; stack.append(r); onstack.add(id(r))
                work.append((r, iter([q for q in gc.get_referents(r) if id(q) in ids])))
                advanced = True
                break
            elif id(r) in onstack:
                low[id(o)] = min(low[id(o)], index[id(r)])
        if advanced:
            continue
        work.pop()
        if work:
            low[id(work[-1][0])] = min(low[id(work[-1][0])], low[id(o)])
        if low[id(o)] == index[id(o)]:
            comp = []
            while True:
                x = stack.pop(); onstack.discard(id(x)); comp.append(x)
                if x is o:
                    break
            if len(comp) > 1:
                comps.append(comp)
print("cycles (strongly connected components):", len(comps))
for comp in sorted(comps, key=len, reverse=True)[:12]:
    print(f"  component of {len(comp)}:")
    for x in comp[:14]:
        print("      ", name(x))
