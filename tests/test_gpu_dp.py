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
for dense_above, mode in ((2.0, "sparse"), (0.0, "dense")):
    step = FineStep(m, process_group=dist.group.WORLD)
    step._sync = GridGradSync(dist.group.WORLD, dense_above=dense_above)
    loss, g = step.forward_loss_backward(local, sc.s_val, global_rays=n, entropy_owner=(rank == world - 1))
    torch.cuda.synchronize()
    assert step._sync.last["mode"] == mode, step._sync.last
    g = {k: v.clone() for k, v in g.items()}
    loss = float(loss)
    ref_loss, ref = FineStep(m).forward_loss_backward(full, sc.s_val)
    torch.cuda.synchronize()
    assert abs(loss - float(ref_loss)) < 1e-6, (loss, float(ref_loss))
    for k, v in ref.items():
        e = float((g[k] - v).abs().max() / v.abs().max().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-5, (mode, k, e)
dist.barrier()
if rank == 0:
    print("DPGPU", worst)
dist.destroy_process_group()
'''


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


def test_bench_flow_with_two_ranks_rehearsal():
    """bench.py's N > 1 flow (launch under torch.distributed.run, barriers, max-over-ranks time, gradient exchange, the
    separately timed optimizer / TV sections on every rank, one JSON line from rank 0) with two real ranks sharing the
    test box's single GPU over gloo (ESR_BENCH_REHEARSAL=1; RCCL refuses two ranks per device)."""
    import json
    env = dict(os.environ, ESR_BENCH_REHEARSAL="1", OMP_NUM_THREADS="2")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
         "--config", "small"],
        env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["grad_exchange"]["mode"] == "sparse" and d["grad_exchange"]["bricks"] > 0      # 2 ranks -> sparse
    assert "REHEARSAL" in d["data"]
