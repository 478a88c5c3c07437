"""Training THROUGH the progressive up-scaling event with WARM step objects (reference: app/fine/fine.py:337-344 -- at
``pg_scale`` steps the grids are re-allocated at the new resolution and the optimizer is rebuilt; cfg/app/fine.yaml:41-43:
160^3 -> 256^3 at step 15000).  The trainer-step objects keep pointer-keyed caches between steps (the engine's pack
cache, the march cache sized by the scene's step bound, the flat gradient buffer, the workspace): after the event every
one of them must follow the new tensors.  k steps at [160,160,40] -> scale_volume_grid -> rebuilt optimizer -> the FIRST
post-event step of the SAME FineStep object against the CPU oracle on the scaled parameters -> k more steps.
Also: the reference's default batch (8192 rays, cfg/app/fine.yaml:51) on the production-size 256^3 grid with oblique
rays, through a size-independent property (a batch's gradient is the sum of its halves' gradients)."""
import dataclasses

import pytest
import torch

from conftest import rel_err
from test_gpu_fine_path import build_gpu_model, gpu_batch, oracle_for, run_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4      # BASELINE.json north_star: 1e-4 rel (to max-norm) fp32
LRS = dict(off_color=0.1, off_rgbnet=0.003, emo_color=0.1, emo_rgbnet=0.003, sdf=0.005, tonemapper=0.003)


def _train(step, opt, b, s_val, k):
    losses = []
    for _ in range(k):
        loss, g = step.forward_loss_backward(b, s_val)
        step.assign_grads(g)
        opt.step()
        losses.append(float(loss))
    return losses


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fine_step_through_scale_volume_grid_event(dtype):
    from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    s_val = 40.0
    sc = slab_scene("g160", s_val=s_val, oblique=True, n_rays=320, seed=5, mask="prune")
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    m.mlp_dtype = dtype
    assert [int(v) for v in m.world_size] == [160, 160, 40]
    b = gpu_batch(sc)
    step = FineStep(m)
    opt = create_optimizer_or_freeze_model(m, **LRS)
    before = _train(step, opt, b, s_val, 3)
    eng = m.engine
    ws_before, cache_before = eng.ws.cap_tiles, {k: v[0] for k, v in eng._pack_cache.items()}
    old_ptrs = {k: p.data_ptr() for k, p in m.named_parameters()}

    # ---- the event (fine.py:337-344): new grids, new optimizer; the step object stays
    scaled = 256 * 256 * 64
    m.scale_volume_grid(scaled)
    assert [int(v) for v in m.world_size] == [256, 256, 64] == m._world_size_l
    assert m.sdf.grid.data_ptr() != old_ptrs["sdf.grid"] and tuple(m.off_color.grid.shape[2:]) == (256, 256, 64)
    assert m.off_color.grid.is_contiguous(memory_format=torch.channels_last_3d)
    opt = create_optimizer_or_freeze_model(m, **LRS)

    # ---- first post-event step, warm objects, against the oracle on the scaled parameters
    sc2 = dataclasses.replace(sc, num_voxels=scaled)
    loss_w, g_w = step.forward_loss_backward(b, s_val)
    torch.cuda.synchronize()
    loss_w, g_w = float(loss_w), {k: v.clone() for k, v in g_w.items()}
    assert tuple(g_w["sdf.grid"].shape[2:]) == (256, 256, 64) and tuple(g_w["off_color.grid"].shape[1:]) == (6, 256, 256, 64)
    assert step._flat.numel() >= 13 * scaled                         # the flat gradient buffer followed the new grids
    if dtype == "f32":
        # the oracle takes over the step's discrete decisions (survivor set, ReLU branches), each arbitrated in float64
        # (tests/decisions.py); rounds 3-4 dropped the rays with a sample on a ReLU kink and re-ran both sides
        from decisions import assert_legitimate, hip_decisions
        dec = hip_decisions(m)
        fp, c, P = oracle_for(m, sc2)
        fp.FLIP_LOG = []
        try:
            o_out, o_loss, o_grads, keep = run_oracle(fp, c, P, sc2, s_val, force=dec)
            assert_legitimate(keep, fp.FLIP_LOG, what="scale event")
        finally:
            fp.FLIP_LOG = None
        lc = m.last_counts
        assert (lc["m0"], lc["m1"], lc["m2"], lc["m3"]) == tuple(keep["counts"])
        assert lc["m0"] > lc["m1"] >= lc["m2"] >= lc["m3"] > 0
        assert abs(loss_w - o_loss) < 1e-5 * max(1.0, abs(o_loss)), (loss_w, o_loss)
        bad = {k: rel_err(g_w[k], g) for k, g in o_grads.items() if not rel_err(g_w[k], g) < TOL}
        assert not bad, bad
        assert set(o_grads) <= set(g_w)
    else:
        # bf16 MLP operands: the same event against the f32 engine on the same (scaled) parameters
        m.mlp_dtype = "f32"
        loss_f, g_f = FineStep(m).forward_loss_backward(b, s_val)
        torch.cuda.synchronize()
        m.mlp_dtype = "bf16"
        assert abs(loss_w - float(loss_f)) < 2e-3 * max(1.0, abs(float(loss_f)))
        for k in ("sdf.grid", "off_color.grid", "emo_color.grid"):
            a, r = g_w[k].flatten().double(), g_f[k].flatten().double()
            cos = float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-30))
            assert cos > 0.995, (k, cos)

    # ---- the caches moved with the tensors
    for which, key in eng._pack_cache.items():
        assert key[0] == cache_before[which]      # MLP parameters are NOT re-allocated by the event: same cache entries
    assert eng.ws.cap_tiles >= ws_before

    # ---- and training goes on: k more steps with the rebuilt optimizer (moments of the new shapes), loss stays finite
    after = _train(step, opt, b, s_val, 3)
    assert all(map(lambda v: v == v and abs(v) < 1e3, before + after))
    for grp in opt.param_groups:
        for p in grp["params"]:
            st = opt.state[p]
            assert tuple(st["exp_avg"].shape) == tuple(p.shape)
    assert after[-1] < after[0]                   # the same batch three times: the loss goes down


def test_default_batch_8192_oblique_on_the_production_grid():
    """cfg/app/fine.yaml:51 (batch 8192) on `C2g256` with tilted rays: one step on the whole batch equals the two
    half-batch steps combined -- survivor counts add up, losses average, gradients add (every term of the trainer loss
    is a mean over rays except the last-ray entropy term, owned by the second half)."""
    from esr_nerf_amd.synthetic import slab_scene
    from esr_nerf_amd.trainer import FineStep
    s_val = 40.0
    sc = slab_scene("C2g256", s_val=s_val, n_rays=8192, oblique=True, seed=11)
    m = build_gpu_model(sc, seed=1, grid_seed=2)
    b = gpu_batch(sc)
    step = FineStep(m)
    loss, g = step.forward_loss_backward(b, s_val)
    torch.cuda.synchronize()
    loss, g, full = float(loss), {k: v.clone() for k, v in g.items()}, dict(m.last_counts)
    assert full["m0"] > full["m1"] >= full["m2"] >= full["m3"] > 8192 * 30
    halves, counts = [], []
    for i, owner in ((0, False), (1, True)):
        bh = {k: v[i * 4096:(i + 1) * 4096].contiguous() for k, v in b.items()}
        lh, gh = step.forward_loss_backward(bh, s_val, global_rays=8192, entropy_owner=owner)
        torch.cuda.synchronize()
        halves.append((float(lh), {k: v.clone() for k, v in gh.items()}))
        counts.append(dict(m.last_counts))
    for k in ("m0", "m1", "m2", "m3"):
        assert counts[0][k] + counts[1][k] == full[k], k
    assert abs(halves[0][0] + halves[1][0] - loss) < 1e-6 * max(1.0, abs(loss))
    for k, v in g.items():
        assert rel_err(halves[0][1][k] + halves[1][1][k], v) < 1e-5, k
