"""Does a training experiment give its device memory back?  (GPU box)  python tools/debug/leak_probe.py [stage]"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import teacher_student as ts
stage = sys.argv[1] if len(sys.argv) > 1 else "pdra"
run = dict(fine=ts.fine_experiment, pdra=ts.pdra_experiment, finetune=ts.finetune_experiment)[stage]
gc.disable()
for seed in range(4):
    run("f32" if seed % 2 == 0 else "bf16", steps=10, seed=seed)
    a = torch.cuda.memory_allocated() / 2**20
    n = gc.collect()
    b = torch.cuda.memory_allocated() / 2**20
    print(f"seed {seed}: allocated {a:.0f} MiB, after gc.collect() ({n} objects) {b:.0f} MiB", flush=True)
from esr_nerf_amd.fine_engine import FineEngine
left = [o for o in gc.get_objects() if isinstance(o, FineEngine)]
print("engines alive:", len(left))
for e in left[:2]:
    for r in gc.get_referrers(e):
        if r is left: continue
        print("  referrer:", type(r).__name__, (list(r.keys())[:8] if isinstance(r, dict) else str(r)[:120]))
big = sorted((o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda), key=lambda t: -t.numel() * t.element_size())[:8]
for t in big:
    print("  tensor", tuple(t.shape), t.dtype, [type(r).__name__ for r in gc.get_referrers(t)][:4])
