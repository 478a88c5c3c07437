"""Fixed cost vs per-tile cost of the weight-gradient launches (GPU box): esr_mlp_wgrad over growing tile ranges."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esr_nerf_amd import _lib
from esr_nerf_amd.fine_engine import FineEngine
eng = FineEngine("cuda:0"); L = eng.L; s = _lib.stream_ptr("cuda:0")
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hid, xrows, nl, zrows, ind, out = (192, 104, 4, 4, 85, 3) if kind == 0 else (192, 48, 2, 4, 33, 3)
T = 16384
X = torch.randn(T, xrows, 32, device="cuda")
H = [torch.randn(T, hid, 32, device="cuda") for _ in range(nl - 1)]
dZ = [torch.randn(T, hid, 32, device="cuda") for _ in range(nl - 1)]
dz = torch.randn(T, zrows, 32, device="cuda")
dims = [ind] + [hid] * (nl - 1) + [out]
gw = [torch.zeros(dims[i + 1], dims[i], device="cuda") for i in range(nl)]
gb = [torch.zeros(dims[i + 1], device="cuda") for i in range(nl)]
for tiles in (256, 512, 1024, 2048, 4096, 8192, 16384):
    def run():
        _lib.check(L.esr_mlp_wgrad(kind, _lib.ptr(X), 0, _lib.ptr_array(H), _lib.ptr_array(dZ), _lib.ptr(dz), 0, tiles,
                                   _lib.ptr_array(gw), _lib.ptr_array(gb), _lib.ptr(eng.wgrad_scratch),
                                   C.c_int64(eng.wgrad_scratch.numel()), s), "wgrad")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"kind {kind} tiles {tiles:6d}  {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per call  ({tiles // 256} tiles per workgroup)")
