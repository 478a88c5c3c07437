"""Fused Adam (esr_adam_step / esr_nerf_amd.optimizer) against a line-by-line torch restatement of the
reference's `adam` (app/utils/optimizer.py:183-228), and a short end-to-end training run."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def reference_adam(p, g, m, v, step, lr, b1=0.9, b2=0.99, eps=1e-8, per_lr=None):
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m * per_lr if per_lr is not None else m, denom, value=-(lr / bc1))


@pytest.mark.parametrize("shape,use_plr", [((1, 1, 17, 9, 5), True), ((192, 85), False), ((1, 6, 16, 12, 10), False),
                                           ((3,), False)])
def test_fused_adam_matches_reference_update(shape, use_plr):
    from esr_nerf_amd.optimizer import Adam
    g = torch.Generator().manual_seed(len(shape))
    p0 = torch.randn(shape, generator=g)
    pr, m, v = p0.clone(), torch.zeros(shape), torch.zeros(shape)
    pd = torch.nn.Parameter(p0.clone().cuda())
    if len(shape) == 5 and shape[1] > 1:
        pd = torch.nn.Parameter(p0.clone().cuda().contiguous(memory_format=torch.channels_last_3d))
    opt = Adam([{"params": [pd], "lr": 0.1, "name": "x"}], betas=(0.9, 0.99))
    plr = None
    if use_plr:
        cnt = torch.randint(0, 50, shape, generator=g)
        opt.set_pervoxel_lr(cnt.cuda())
        plr = cnt.float() / cnt.max()
    for step in range(1, 8):
        grad = torch.randn(shape, generator=g) * (0.0 if step == 4 else 1.0)     # a zero-gradient step too
        grad[..., 0] = 0.0
        reference_adam(pr, grad, m, v, step, 0.1, per_lr=plr)
        pd.grad = grad.cuda()
        opt.step()
        assert rel_err(pd, pr) < 2e-6, step
    st = opt.state[pd]
    assert st["step"] == 7 and rel_err(st["exp_avg"], m) < 2e-6 and rel_err(st["exp_avg_sq"], v) < 2e-6
    assert opt.name2pg["x"]["lr"] == 0.1


def test_short_training_run_reduces_loss_and_tracks_torch_adam():
    """20 steps of FineStep + fused Adam on a slab scene: the loss goes down, and the trajectory stays
    on top of the same 20 steps taken with torch.optim.Adam on the autograd route."""
    from esr_nerf_amd.config import fine_cfg
    from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
    from esr_nerf_amd.synthetic import init_slab_model, slab_scene
    from esr_nerf_amd.trainer import FineStep
    from esr_nerf_amd.voxurff import VoxurfF
    from oracle import fine_path as fp
    sc = slab_scene("small", s_val=30.0, n_rays=512, oblique=True, seed=4)
    lrs = dict(off_color=0.1, off_rgbnet=0.003, emo_color=0.1, emo_rgbnet=0.003, sdf=0.005, tonemapper=0.003)

    def build():
        torch.manual_seed(0)
        np.random.seed(0)
        m = VoxurfF(fine_cfg("cuda:0"), sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.mask_xyz_min, sc.mask_xyz_max,
                    sc.mask_alpha_init, sc.mask_density, sc.s_val, sc.num_voxels)
        init_slab_model(m, sc, seed=1)
        m.train()
        return m
    b = {k: v.cuda() for k, v in sc.batch.items()}
    m1 = build()
    opt1 = create_optimizer_or_freeze_model(m1, **lrs)
    step = FineStep(m1)
    losses = []
    for _ in range(20):
        loss, grads = step.forward_loss_backward(b, 30.0)
        step.assign_grads(grads)
        opt1.step()
        losses.append(float(loss))
    assert losses[-1] < 0.8 * losses[0], losses
    m2 = build()
    groups = [{"params": list(getattr(m2, k).parameters()), "lr": lr} for k, lr in lrs.items()]
    opt2 = torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-8)
    l2 = []
    for _ in range(20):
        opt2.zero_grad(set_to_none=True)
        res = m2(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"], s_val=30.0)
        loss, _ = fp.fine_loss(res, b["rgbs"])
        loss.backward()
        opt2.step()
        l2.append(float(loss))
    assert max(abs(a - c) for a, c in zip(losses, l2)) < 1e-2 * max(losses), (losses, l2)   # 20 steps amplify fp32 atomics-order noise
