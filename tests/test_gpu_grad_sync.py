"""Brick kernels of the data-parallel gradient exchange (csrc/brick.hip) against their torch restatement,
bit-exact (pure data movement), and GridGradSync end to end on a one-rank process group."""
import os

import pytest
import torch

from brick_ops_double import TorchBrickOps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _close_process_group():
    yield
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [128 * 7, 128 * 1000 + 37, 5, 128 * 4096])
def test_brick_kernels_match_torch(n):
    from esr_nerf_amd.grad_sync import HipBrickOps
    hip, ref = HipBrickOps(), TorchBrickOps()
    assert hip.brick == ref.brick
    g = torch.Generator().manual_seed(n)
    flat = torch.randn(n, generator=g)
    nb = (n + 127) // 128
    keep = torch.rand(nb, generator=g) < 0.3
    keep[-1] = True                                             # exercise the ragged last brick
    flat = (flat.new_zeros(nb * 128).view(nb, 128) + keep[:, None]).view(-1)[:n] * flat
    flat[3] = 0.0
    d = flat.cuda()
    f_hip = torch.empty(nb, dtype=torch.uint8, device="cuda")
    f_ref = torch.empty(nb, dtype=torch.uint8)
    hip.flags(d, f_hip)
    ref.flags(flat, f_ref)
    assert torch.equal(f_hip.cpu(), f_ref)
    idx = f_ref.nonzero().view(-1)
    p_hip = torch.full((idx.numel() * 128,), 7.0, device="cuda")
    p_ref = torch.empty(idx.numel() * 128)
    hip.pack(d, idx.cuda(), p_hip)
    ref.pack(flat, idx, p_ref)
    assert torch.equal(p_hip.cpu(), p_ref)
    out = torch.full((n,), -1.0, device="cuda")
    hip.unpack(p_hip * 2, idx.cuda(), out)
    expect = torch.full((n,), -1.0)
    ref.unpack(p_ref * 2, idx, expect)
    assert torch.equal(out.cpu(), expect)


def test_brick_ops_refuse_cpu_tensors():
    from esr_nerf_amd.grad_sync import HipBrickOps
    with pytest.raises(RuntimeError, match="device tensors"):
        HipBrickOps().flags(torch.zeros(256), torch.zeros(2, dtype=torch.uint8))


@pytest.mark.parametrize("density,mode", [(0.2, "sparse"), (0.95, "dense"), (0.0, "sparse")])
def test_grid_grad_sync_one_rank_group(density, mode):
    """World size 1 over RCCL: the exchange must leave the buffer unchanged whichever branch it takes."""
    import torch.distributed as dist
    from esr_nerf_amd.grad_sync import GridGradSync
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1)
    nb = 5000
    g = torch.Generator().manual_seed(3)
    keep = (torch.rand(nb, generator=g) < density).float()
    flat = (torch.randn(nb, 128, generator=g) * keep[:, None]).view(-1).cuda()
    want = flat.clone()
    sync = GridGradSync(dist.group.WORLD)
    sync.reduce(flat)
    torch.cuda.synchronize()
    assert torch.equal(flat, want)
    assert sync.last["mode"] == mode and sync.last["bricks"] == nb
    if mode == "sparse":
        assert sync.last["sent"] == int(keep.sum())


@pytest.mark.parametrize("nb,density,cap", [(5000, 0.2, 1300), (5000, 0.2, 700), (37, 0.5, 64), (426_000, 0.16, 90_000),
                                            (1_705_000, 0.05, 100_000), (300, 0.0, 16), (1, 1.0, 4)])
def test_brick_list_matches_torch(nb, density, cap):
    """esr_brick_list (the union's fixed-capacity brick list + count, one device call) against the torch form it
    replaces: the first `cap` flagged bricks in ascending order, -1 in the unused slots, the true count even when it
    exceeds the capacity."""
    from esr_nerf_amd.grad_sync import HipBrickOps
    g = torch.Generator().manual_seed(nb + cap)
    flags = (torch.rand(nb, generator=g) < density).to(torch.uint8)
    want = flags.nonzero().view(-1)
    idx, count = HipBrickOps().list(flags.cuda(), cap)
    torch.cuda.synchronize()
    assert int(count) == want.numel()
    k = min(cap, want.numel())
    assert torch.equal(idx[:k].cpu(), want[:k]) and bool((idx[k:] == -1).all()) and idx.numel() == cap


def test_grid_grad_sync_optimistic_steps_and_overflow_on_device_list():
    """Steps 2+ of GridGradSync run on the device-built list (no host wait inside); a union that outgrows the capacity
    is completed by verify()'s second pass.  One-rank RCCL group: the buffer must come back unchanged every time."""
    import torch.distributed as dist
    from esr_nerf_amd.grad_sync import GridGradSync
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1)
    nb = 20000
    g = torch.Generator().manual_seed(5)
    sync = GridGradSync(dist.group.WORLD)
    for dens in (0.10, 0.11, 0.30, 0.30, 0.05):                # step 3 overflows the capacity learnt from step 2
        keep = (torch.rand(nb, generator=g) < dens).float()
        flat = (torch.randn(nb, 128, generator=g) * keep[:, None]).view(-1).cuda()
        want = flat.clone()
        sync.reduce(flat)
        sync.verify()
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        assert sync.last["union"] == int(keep.sum())
    ph = sync.profile(flat)
    assert set(ph) >= {"flags", "list", "pack", "unpack", "allreduce_packed"} and torch.equal(flat, want)
