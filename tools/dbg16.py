import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fine_path as T
orig = T.rel_err
calls = []
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    per_tile = ((a - b).abs().flatten(1).max(1).values / b.abs().max()).tolist() if a.dim() == 3 else None
    print("rel_err", round(e, 5), "worst tiles", sorted([(round(x, 4), i) for i, x in enumerate(per_tile)])[-4:] if per_tile else None)
    if a.dim() == 3 and e > 1e-2:
        wt = int((a - b).abs().flatten(1).max(1).values.argmax())
        d = (a - b).abs()[wt] / b.abs().max()
        print("  worst tile", wt, "rows with err:", (d.max(1).values > 4e-3).nonzero().flatten().tolist()[:40])
        print("  cols with err:", (d.max(0).values > 4e-3).nonzero().flatten().tolist()[:40])
    return 0.0
T.rel_err = rel
import torch as _t
_orig_zeros = _t.zeros
keepH = []
def _z(*a, **k):
    t = _orig_zeros(*a, **k)
    if len(a) == 3 and a[2] == 32 and k.get('device') == 'cuda': keepH.append(t)
    return t
_t.zeros = _z
T.test_mlp_engine_bf16_vs_emulation(int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[3]) if len(sys.argv) > 3 else 0)

