"""Two real ranks through the product data-parallel step (FineStep + GridGradSync + the HIP brick kernels):
both processes share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device);
the reduced shard gradients must equal the single-process gradients of the full batch."""
import os
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from esr_nerf_amd.config import fine_cfg
from esr_nerf_amd.grad_sync import GridGradSync
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import FineStep, shard_batch
from esr_nerf_amd.voxurff import VoxurfF
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
sc = slab_scene("tiny", s_val=40.0, oblique=True)
torch.manual_seed(0); np.random.seed(0)
m = VoxurfF(fine_cfg("cuda:0"), sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
            sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc)
m.train()
full = {k: v.cuda() for k, v in sc.batch.items()}
n = full["rays_o"].shape[0]
local = shard_batch(full, rank, world)
worst = 0.0
ref_loss, ref = FineStep(m).forward_loss_backward(full, sc.s_val)
ref = {k: v.clone() for k, v in ref.items()}
torch.cuda.synchronize()
for dense_above, mode in ((2.0, "sparse"), (0.0, "dense")):
    step = FineStep(m, process_group=dist.group.WORLD)
    step._sync = GridGradSync(dist.group.WORLD, dense_above=dense_above, min_capacity=1)
    # three steps: the first sizes the brick list with one exact pass, the later ones use the host-known capacity
    # (no wait inside the exchange; HIP pack / unpack with unused -1 slots) -- all must give the full-batch gradients
    for it in range(3):
        loss, g = step.forward_loss_backward(local, sc.s_val, global_rays=n, entropy_owner=(rank == world - 1))
        torch.cuda.synchronize()
        assert step._sync.last["mode"] == mode, step._sync.last
        assert abs(float(loss) - float(ref_loss)) < 1e-6, (float(loss), float(ref_loss))
        for k, v in ref.items():
            e = float((g[k] - v).abs().max() / v.abs().max().clamp_min(1e-30))
            worst = max(worst, e)
            assert e < 1e-5, (mode, it, k, e)
    if mode == "sparse":
        assert step._sync.last["sent"] >= step._sync.last["union"] > 0 and "overflow" not in step._sync.last
    step.close()                                     # the last step's deferred march-overflow flag: nothing flagged

# the split-fp16 kernels' range fallback is RANK-LOCAL and decided before the exchange: rank 1's flag is raised (as a forward
# launch of its shard would raise it), rank 1 re-runs its step on the f32 MFMA kernels, rank 0 does not -- both join ONE
# exchange, and every rank ends with the full-batch gradients
import warnings
for mode in ("sparse", "dense"):
    step = FineStep(m, process_group=dist.group.WORLD)
    step._sync_mode = mode
    step.forward_loss_backward(local, sc.s_val, global_rays=n, entropy_owner=(rank == world - 1))
    before = m.engine.split_fallback_steps
    if rank == 1:
        m.engine.range_flag.fill_(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        loss, g = step.forward_loss_backward(local, sc.s_val, global_rays=n, entropy_owner=(rank == world - 1))
    torch.cuda.synchronize()
    assert m.engine.split_fallback_steps - before == (1 if rank == 1 else 0), (rank, m.engine.split_fallback_steps, before)
    assert int(m.engine.range_flag) == 0 and m.engine.split_fwd
    assert abs(float(loss) - float(ref_loss)) < 1e-5, (float(loss), float(ref_loss))
    for k, v in ref.items():
        e = float((g[k] - v).abs().max() / v.abs().max().clamp_min(1e-30))
        assert e < 2e-5, ("range fallback on one rank", mode, k, e)
    step.close()

# a march overflow in the LAST step of a data-parallel run must not be dropped: only rank 1's rays overflow (its scene
# gets a step bound of 4 -> an LDS capacity of 64 steps, the tilted rays of this scene take up to ~68), the flag travels
# with the loss all-reduce, and close() raises on EVERY rank
sc2 = slab_scene("small", s_val=40.0, oblique=True)
m2 = VoxurfF(fine_cfg("cuda:0"), sc2.near, sc2.far, sc2.xyz_min, sc2.xyz_max, sc2.mask_xyz_min, sc2.mask_xyz_max,
             sc2.mask_alpha_init, sc2.mask_density, sc2.s_val, sc2.num_voxels)
init_slab_model(m2, sc2)
m2.train()
full2 = {k: v.cuda() for k, v in sc2.batch.items()}
local2 = shard_batch(full2, rank, world)
n2 = full2["rays_o"].shape[0]
real_scene = m2.scene_struct
def tight():
    sc_ = real_scene()
    if rank == 1:
        sc_.max_steps = 4
    return sc_
with FineStep(m2, process_group=dist.group.WORLD) as step:         # (context manager: close() at exit, nothing flagged)
    step.forward_loss_backward(local2, sc2.s_val, global_rays=n2, entropy_owner=(rank == world - 1))
step = FineStep(m2, process_group=dist.group.WORLD)
step.forward_loss_backward(local2, sc2.s_val, global_rays=n2, entropy_owner=(rank == world - 1))
m2.scene_struct = tight
step.forward_loss_backward(local2, sc2.s_val, global_rays=n2, entropy_owner=(rank == world - 1))     # raises nothing yet
m2.scene_struct = real_scene
raised = False
try:
    step.close()
except RuntimeError as e:
    raised = "max_steps" in str(e)
assert raised, "close() did not report the last step's overflow on rank %d" % rank
del m2, step

# ---- option 2 (ESR_GRAD_SYNC=shard): reduce-scatter + Adam on the owned shard + all-gather.  Two checks per step:
# (a) the reduce-scattered gradient, gathered back, equals the full-batch gradient (1e-5 of each grid's largest);
# (b) the sharded update equals the dense fused Adam fed with exactly those gradient values (elementwise the same
#     kernel: agreement to rounding) -- shard boundaries, learning rate by position, moments, parameter all-gather.
# (Comparing the updates of two separately summed gradients instead would test Adam's sensitivity to summation order:
#  an element whose gradient cancels to rounding noise moves by up to lr either way.)
from esr_nerf_amd.grad_sync import ShardedGrids, _all_gather
from esr_nerf_amd.optimizer import Adam, ShardedGridAdam
names = ["sdf", "off_color", "emo_color"]
lrs = {"sdf": 0.005, "off_color": 0.1, "emo_color": 0.1}
start = {k: getattr(m, k).grid.detach().clone() for k in names}
dense_p = {k: torch.nn.Parameter(v.clone()) for k, v in start.items()}
dense_opt = Adam([{"params": [dense_p[k]], "lr": lrs[k], "name": k} for k in names], betas=(0.9, 0.99))
step = FineStep(m, process_group=dist.group.WORLD)
step.sharded = ShardedGrids(m, names, dist.group.WORLD)
SG = step.sharded
assert m.off_color.grid.is_contiguous(memory_format=torch.channels_last_3d)
opt = ShardedGridAdam(SG, lrs)
for it in range(3):
    _, gf = FineStep(m).forward_loss_backward(full, sc.s_val)                 # full batch, same parameters
    gf = {k: gf[k + ".grid"].clone() for k in names}
    loss, g = step.forward_loss_backward(local, sc.s_val, global_rays=n, entropy_owner=(rank == world - 1))
    SG.wait()
    gsum = torch.zeros(SG.padded, device="cuda")
    _all_gather(gsum, SG.grad_shard.clone(), dist.group.WORLD)
    torch.cuda.synchronize()
    for k, (name, b0, b1) in zip(names, SG.bounds):
        p = dense_p[k]
        seg = gsum[b0:b1]
        gk = seg.view(p.shape[0], *p.shape[2:], p.shape[1]).permute(0, 4, 1, 2, 3) if p.shape[1] > 1 else seg.view(p.shape)
        e = float((gk - gf[k]).abs().max() / gf[k].abs().max().clamp_min(1e-30))
        assert e < 1e-5, ("shard-grad", it, k, e)
        p.grad = gk.clone()
    opt.step()
    dense_opt.step()
    torch.cuda.synchronize()
    for k in names:
        cur = getattr(m, k).grid.detach()
        e = float((cur - dense_p[k].detach()).abs().max())
        assert e < 1e-3 * lrs[k], ("shard-adam", it, k, e)
        assert float((cur - start[k]).abs().max()) > 0.5 * lrs[k]             # ... and the update did happen
dist.barrier()
if rank == 0:
    print("DPGPU", worst)
dist.destroy_process_group()
'''


def _why(out):
    """The lines of a failed worker's output that say why (pytest cuts long assertion messages in the middle)."""
    keep = [l for l in (out.stdout + "\n" + out.stderr).splitlines()
            if any(k in l for k in ("Error", "error", "assert", "Traceback", "File \"", "DPLTS", "DPGPU", "killed", "Signal", "exitcode"))
            and "elastic/errors" not in l]
    return "\n".join(keep[-40:])


LTS_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from esr_nerf_amd.config import lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from esr_nerf_amd.trainer import LtsStep, lts_point_share, shard_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
s_val = 60.0
sc = slab_scene("small", s_val=s_val, oblique=True, n_rays=384, seed=2)
torch.manual_seed(0); np.random.seed(0)
cfg = lts_cfg("cuda:0", num_2ndrays=16, num_ltspts=25)      # (odd: the split leaves a remainder)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
init_slab_model(m, sc, seed=3)
with torch.no_grad():
    m.brdf.grid.normal_(0.0, 0.3, generator=torch.Generator(device="cuda").manual_seed(5))
m.train()
full = {k: v.cuda() for k, v in sc.batch.items()}
full["uncert_masks"] = (torch.arange(384, device="cuda") % 3 == 0)
n = 384
worst = 0.0
for stage in ("lts", "pdra"):
    m.pdra_mode = stage == "pdra"
    for split_points in (False, True):
        # the data-parallel step: every rank its ray shard and its own draws (seeded per rank)
        step = LtsStep(m, cfg.app.trainer, stage=stage, process_group=dist.group.WORLD, split_points=split_points)
        if split_points:                                 # the ranks' shares add up to the reference's number of points
            cnt = torch.tensor([step.ltspts])
            dist.all_reduce(cnt)
            assert int(cnt) == m.num_ltspts == 25 and step.ltspts == (13 if rank == 0 else 12)
        torch.manual_seed(100 + rank); np.random.seed(100 + rank)
        loss, G, _ = step.forward_loss_backward(shard_batch(full, rank, world), s_val, global_rays=n, entropy_owner=(rank == world - 1))
        torch.cuda.synchronize()
        loss, G = float(loss), {k: v.clone() for k, v in G.items()}
        step.close()
        # what it must equal: the SUM over the shards of the single-process steps on each shard with the same draws and the
        # same global normalisation (every rank computes all of them locally)
        ref_loss, ref = 0.0, None
        for r in range(world):
            one = LtsStep(m, cfg.app.trainer, stage=stage)
            if split_points:
                one.ltspts, one.pt_scale = lts_point_share(m.num_ltspts, world, r)
            torch.manual_seed(100 + r); np.random.seed(100 + r)
            l_r, G_r, _ = one.forward_loss_backward(shard_batch(full, r, world), s_val, global_rays=n, entropy_owner=(r == world - 1))
            torch.cuda.synchronize()
            ref_loss += float(l_r)
            ref = {k: v.clone() for k, v in G_r.items()} if ref is None else {k: ref[k] + G_r[k] for k in ref}
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (stage, split_points, loss, ref_loss)
        assert set(G) == set(ref) and len(G) == 43
        for k, v in ref.items():
            e = float((G[k] - v).abs().max() / v.abs().max().clamp_min(1e-30))
            worst = max(worst, e)
            assert e < 2e-5, (stage, split_points, k, e)
dist.barrier()
if rank == 0:
    print("DPLTS", worst)
dist.destroy_process_group()
'''


def test_two_rank_lts_step_equals_the_sum_of_its_shards():
    """LtsStep / PDRA under data parallelism: two real ranks on the one GPU over gloo, each with its ray shard and
    its own random draws (surface points, directions, noises: seeded per rank), with and without ``split_points``.  The
    all-reduced loss and all 43 gradients equal the sum over the shards of the single-process steps on each shard with the same
    draws and the same global normalisation -- the exchange adds nothing and loses nothing."""
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "w.py")
        open(w, "w").write(LTS_WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
        out = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
             "--master-addr", "127.0.0.1", "--master-port", "29541", w, ROOT],
            env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, _why(out)
    line = [l for l in out.stdout.splitlines() if l.startswith("DPLTS")][0].split()
    assert float(line[1]) < 2e-5


def test_two_rank_step_equals_full_batch_step():
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "w.py")
        open(w, "w").write(WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
        out = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
             "--master-addr", "127.0.0.1", "--master-port", "29533", w, ROOT],
            env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("DPGPU")][0].split()
    assert float(line[1]) < 1e-5


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_flow_with_two_ranks_rehearsal(scaling):
    """bench.py's N > 1 flow (launch under torch.distributed.run, barriers, max-over-ranks time, gradient exchange, the
    separately timed optimizer / TV sections on every rank, one JSON line from rank 0) with two real ranks sharing the
    test box's single GPU over gloo (ESR_BENCH_REHEARSAL=1; RCCL refuses two ranks per device)."""
    import json
    env = dict(os.environ, ESR_BENCH_REHEARSAL="1", OMP_NUM_THREADS="2")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
         "--config", "small", "--scaling", scaling],
        env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["rays_per_gpu"] == (512 if scaling == "weak" else 256)      # strong: the config's rays in total
    assert d["grad_exchange"]["mode"] == "sparse" and d["grad_exchange"]["bricks"] > 0      # 2 ranks -> sparse
    assert "REHEARSAL" in d["data"]


@pytest.mark.parametrize("scaling,stage", [("weak", "fine"), ("strong", "lts")])
def test_bench_flow_with_the_dense_exchange_rehearsal(scaling, stage):
    """What a node with >= 5 GPUs runs, before a real node sees it: the `dense` exchange form (one asynchronous
    all-reduce under the weight-gradient kernels; the default from five ranks on: trainer.default_grid_sync, asserted for every
    world size on the CPU) and the forced exchange sweep, in bench.py's flow -- weak (fine stage) and strong (lts stage: rays AND
    surface points split) -- with FOUR real ranks on the test box's one GPU over gloo and ESR_GRAD_SYNC=dense.  Four, not eight
    or five: this pool kills a run with more than six processes on a card, the test runner is one of them, and a five-rank run
    inside the full suite was killed with seven counted; the rank arithmetic of eight (point shares, shard sizes, the default's
    choice) is covered on the CPU (tests/test_host.py)."""
    import json
    env = dict(os.environ, ESR_BENCH_REHEARSAL="1", OMP_NUM_THREADS="2", ESR_GRAD_SYNC="dense")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
         "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
         "--config", "small", "--stage", stage, "--scaling", scaling, "--sync-sweep", "--no-cpu-baseline", "--no-other"],
        env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["scaling"] == scaling and d["value"] > 0 and "REHEARSAL" in d["data"]
    assert d["grad_exchange"]["mode"].startswith("dense")                          # one asynchronous all-reduce
    assert set(d["grad_sync_sweep_ms"]) >= {"sparse", "dense"}                     # the sweep ran both forms
    assert d["split_fallback_steps"] == 0


def test_plain_bench_command_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the shape of the driver's single-GPU command): bench.py starts
    `torch.distributed.run` itself as a child process and relays rank 0's JSON line as the last line of stdout, exit code 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ESR_BENCH_REHEARSAL="1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
                          "--config", "small"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    last = out.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["dist"]["rccl_ranks"] == 2
    assert d["split_fallback_steps"] == 0


def test_bench_under_rccl_prints_one_json_line_and_nothing_else_on_stdout():
    """The driver's launcher line with ONE rank on the real backend (`nccl` = RCCL; two ranks on one card are refused by it):
    process-group set-up with `device_id`, the exchange code with a world of one -- and stdout holds rank 0's JSON line only
    (RCCL prints a banner to stdout when its communicator comes up; bench.py points the descriptor at stderr meanwhile)."""
    import json
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
         "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
         "--config", "small", "--force-dist", "--no-cpu-baseline", "--no-other", "--no-optimizer"],
        env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, _why(out)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
