"""Does the step's ramp after a short warm-up (bench.py --step-times) follow the GPU's clock?  (GPU box)
   python tools/debug/clock_ramp.py
Samples the current shader clock from sysfs (pp_dpm_sclk: the line marked '*') in a thread every millisecond while steps run
from an idle device, and prints step end times next to the clock seen at that moment."""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import io, contextlib
import numpy as np, torch
from esr_nerf_amd.config import fine_cfg
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import FineStep
from esr_nerf_amd.voxurff import VoxurfF

MODE = sys.argv[1] if len(sys.argv) > 1 else "sleep"          # sleep | spin (the host busy-waits instead of sleeping) | gpu (a dummy device load)
paths = []


def sclk():
    out = []
    for p in paths + extra:
        try:
            for line in open(p):
                if "*" in line:
                    out.append(line.split(":")[1].strip().rstrip("*").strip())
        except OSError as e:
            out.append(f"({e.__class__.__name__})")
    return "/".join(out)


sc = slab_scene("C2", s_val=20.0)
torch.manual_seed(0); np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = VoxurfF(fine_cfg("cuda:0"), sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc)
m.train()
b = {k: v.cuda() for k, v in sc.batch.items()}
step = FineStep(m)
step.forward_loss_backward(b, 20.0)          # allocations
torch.cuda.synchronize()
# which of our eight cards' files is this process's device: the one whose clock moves when we load it
def all_clocks(kind):
    out = {}
    for p in sorted(glob.glob(f"/sys/class/drm/card*/device/pp_dpm_{kind}")):
        try:
            out[p] = [l.split(":")[1].strip().rstrip("*").strip() for l in open(p) if "*" in l][0]
        except (OSError, IndexError):
            out[p] = "?"
    return out
time.sleep(1.0)
idle = all_clocks("sclk")
x = torch.randn(8192, 8192, device="cuda")
for _ in range(20):
    x = x @ x * 1e-4
torch.cuda.synchronize()
busy = all_clocks("sclk")
mine = [p for p in idle if idle[p] != busy[p]]
print("sclk idle -> busy:", {os.path.basename(os.path.dirname(os.path.dirname(p))): (idle[p], busy[p]) for p in mine})
paths = [p for p in mine[:1]]
extra = [paths[0].replace("sclk", k) for k in ("mclk", "fclk", "socclk")] if paths else []
del x
if MODE == "sleep":
    time.sleep(1.0)                              # an idle device AND an idle host, as after a blocking wait
elif MODE == "spin":
    t_ = time.perf_counter()
    while time.perf_counter() - t_ < 1.0:        # an idle device, a busy host core
        pass
else:
    y = torch.randn(8192, 8192, device="cuda")
    t_ = time.perf_counter()
    while time.perf_counter() - t_ < 1.0:        # a busy device (and host)
        y = y @ y * 1e-4
    torch.cuda.synchronize()
    del y
samples, stop = [], False


def poll():
    while not stop:
        samples.append((time.perf_counter(), sclk()))
        time.sleep(0.001)


th = threading.Thread(target=poll); th.start()
ends = []
t0 = time.perf_counter()
for i in range(40):
    step.forward_loss_backward(b, 20.0)
    torch.cuda.synchronize()
    ends.append(time.perf_counter())
stop = True; th.join()
prev = t0
for i, e in enumerate(ends):
    near = min(samples, key=lambda s: abs(s[0] - e))[1] if samples else "?"
    if i < 14 or i % 6 == 0:
        print(f"[{MODE}] step {i:2d}: {1e3 * (e - prev):6.3f} ms (synchronised)   sclk {near}")
    prev = e
