#!/usr/bin/env python3
"""Benchmark of the fine-stage hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* = one pass of the hot path over one batch of synthetic input: renderer
forward (ray march, feature gather, radiance/tonemap MLPs, compositing), the
trainer's loss, and the full backward -- BASELINE.json config C2 ("giftbox_w fine
stage, 4096 rays x 128 samples, fp32") on the slab scene of SURVEY.md section 8(d),
inputs resident in HBM before the timed region.  The optimizer step is outside
the named path (reported separately, not here).  With N > 1 every rank
runs the same step on its own 4096 rays (weak scaling) and the gradients are
all-reduced over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel, algorithmic FLOPs / HIP-event time vs the f32 MFMA peak
  cpu_baseline  oracle/fine_path.py (the CPU port of the reference path) timed on
                the host cores on a bounded ray sample of the same scene (N=1 only)
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With an RCCL communicator alive its
# streams take queues too, and the engine's side stream (weight gradients beside the grid scatters, packing beside the
# march) landed on the SAME in-order hardware queue as the main stream: everything serialised, +0.12 ms per step
# (tools/trace_step.py on `bench.py --force-dist`: one queue id for all kernels).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
MFMA_F16_PEAK_TF = 2500.0         # same guide, dense FP16 / BF16 MFMA
MFMA_BF16_PEAK_TF = 2500.0        # same guide, dense BF16 MFMA
HBM_PEAK_GBS = 8000.0             # same guide, HBM3E spec
# net -> (inputs, hidden width, hidden layers, outputs, input rows that receive a gradient)    app/utils/pbr/module.py:6-83
NETS = {"off": (85, 192, 3, 3, 43), "emo": (85, 192, 3, 3, 43), "tone": (33, 192, 1, 3, 33),
        "brdf": (76, 128, 3, 5, 43), "emit": (76, 128, 3, 3, 43)}


def net_macs(net, op):
    """Algorithmic MACs per sample of one MLP pass (padding excluded).  fwd and wgrad: every weight once; dgrad: the
    transposed weights, the first layer only towards the grid-fed input rows."""
    i, h, nh, o, gi = NETS[net]
    if op == "dgrad":
        return o * h + (nh - 1) * h * h + h * gi
    return i * h + (nh - 1) * h * h + h * o


def net_bytes(net, op, bf16):
    """Algorithmic HBM bytes per sample of one MLP launch: inputs read + activations / gradients saved for (or read
    by) the other passes.  Hidden tiles are stored as bf16 (2 B) by the bf16 engine, fp32 (4 B) otherwise; the
    network input X, the output z and their gradients are fp32 in both; ReLU masks 1 bit per unit."""
    i, h, nh, o, gi = NETS[net]
    hb = 2 if bf16 else 4
    if op == "fwd":
        return i * 4 + nh * h * hb + nh * h // 8 + 4 * 4
    if op == "dgrad":
        return 4 * 4 + nh * h // 8 + nh * h * hb + gi * 4
    return i * 4 + 4 * 4 + 2 * nh * h * hb          # wgrad: X, dz, H and dZ of every hidden layer


RAD_MAC, TONE_MAC = net_macs("off", "fwd"), net_macs("tone", "fwd")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5", "small", "tiny"],
                    help="BASELINE.json configs: C2 fine fp32 (headline), C3 fine 192 samples, C4 lts, C5 = C4 scene, pdra stage, "
                         "bf16 MLPs")
    ap.add_argument("--stage", default=None, choices=["fine", "lts", "pdra", "finetune"],
                    help="trainer step to run (default: fine; C4 defaults to lts; finetune = the re-lighting fine-tune "
                         "half of C5, pdra.py:1047-1109: forward_finetune + its loss + backward)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = the config's rays PER GPU (default, what the driver runs); strong = the config's "
                         "rays in TOTAL, split over the ranks (SURVEY 8(d) asks for both)")
    ap.add_argument("--s-val", type=float, default=None, help="default 20 (fine.yaml:45) / 220 (lts.yaml:52)")
    ap.add_argument("--oblique", action="store_true",
                    help="tilted, un-normalised ray directions (synthetic.slab_scene(oblique=True)): NOT BASELINE's workload "
                         "(its rays are axis-parallel) -- a robustness line for the kernels that exploit ray coherence")
    ap.add_argument("--grid", type=int, default=None, choices=[256],
                    help="C2 only: the production-size grid of the fine stage's end (cfg/app/fine.yaml:41-43), world "
                         "256 x 256 x 256, same 4096 rays x 128 surviving samples (synthetic.CONFIGS['C2g256']) -- a labelled "
                         "robustness line, NOT BASELINE's workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=None,
                    help="rays of the CPU baseline sample (default: the whole batch in the fine stage -- SURVEY 8(d): same N, "
                         ">= 5 iterations -- and 1024 in the LTS stages, whose CPU step is ~10x longer)")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="MLP operand type: f32 (f32 matrix cores, the headline) or bf16 (bf16 operands, fp32 accumulation)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gradient all-reduces even with one rank (self-test)")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the separately reported fused-Adam timing")
    ap.add_argument("--no-tv", action="store_true",
                    help="leave the trainers' every-third-iteration TV lines (SURVEY A12; fine.py:383-400, lts.py:381-398, pdra.py:459-476) out of "
                         "the timed steps (they are inside by default: the metric is forward A1-A12 + loss + backward)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket kernels with HIP events (drops the roofline object)")
    ap.add_argument("--sync-sweep", action="store_true",
                    help="N > 1: after the timed steps, time 10 more steps under each gradient-exchange form (sparse / dense; "
                         "shard is left out: it needs the sharded optimizer) and print grad_sync_sweep_ms, so that ONE multi-GPU "
                         "record is enough to pick the default of trainer._grid_sync.  On by itself for N = 2..4 (outside the timed "
                         "region, ~0.1 s); for N >= 5 only with this flag: there the non-default form is the one no real node has run yet")
    ap.add_argument("--no-sync-sweep", action="store_true", help="N = 2..4: leave the sweep out")
    ap.add_argument("--step-times", action="store_true",
                    help="diagnostic: record a HIP event behind every timed step and print the per-step times (ms) to stderr -- how "
                         "long the step takes to reach its steady state after the warm-up (clocks, caches)")
    ap.add_argument("--serial", action="store_true",
                    help="every kernel on ONE stream (no weight gradients beside the scatters, no side streams in the LTS steps): "
                         "for rocprofv3 kernel statistics, whose per-kernel times should not include what ran beside them")
    ap.add_argument("--no-other", action="store_true",
                    help="headline run only: skip the `other_workloads` key (C3 bf16, C4 lts f32, C5 pdra bf16; 30 steps each, "
                         "one child process per workload after the headline has been measured)")
    return ap.parse_args()


def fine_calls(counts, merged_off):
    """engine call name -> [(net, op, samples)] of the fine-stage step."""
    n_on, n_off = counts["n_on"], counts["n_off"]
    n_all = n_on + n_off
    c = {"mlp_fwd(off|on-tiles)": [("off", "fwd", n_on)],
         "mlp_fwd(off)": [("off", "fwd", n_all if merged_off else n_off)],
         "mlp_fwd(emo)": [("emo", "fwd", n_on)], "mlp_fwd(tone)": [("tone", "fwd", n_all)],
         # f32 engine, merged launches (esr_mlp_fwd_fine / esr_mlp_dgrad_fine)
         "mlp_fwd(rad)": [("off", "fwd", n_all), ("emo", "fwd", n_on)],
         "mlp_dgrad(rad)": [("emo", "dgrad", n_on), ("off", "dgrad", n_off)],
         "mlp_dgrad(emo)": [("emo", "dgrad", n_on)], "mlp_dgrad(off)": [("off", "dgrad", n_off)],
         "mlp_dgrad(tone)": [("tone", "dgrad", n_all)],
         # f32 engine: the tone mapper's weight gradients are a kernel of their own (csrc/tone_wgrad.hip)
         "mlp_wgrad(all)": [("emo", "wgrad", n_on), ("off", "wgrad", n_off)] + ([] if merged_off else [("tone", "wgrad", n_all)]),
         "tone_wgrad": [("tone", "wgrad", n_all)]}
    return c


def algorithmic_flops(name, counts):
    """Algorithmic FLOPs of one launch of a named MLP call (padding and the detached-pass bookkeeping excluded):
    2 * MACs * samples it processes."""
    calls = fine_calls(counts, counts.get("merged_off_pass", False)).get(name)
    return sum(2 * net_macs(n, op) * k for n, op, k in calls) if calls else None


def algorithmic_bytes(name, counts, bf16=True):
    """Algorithmic HBM bytes of one launch of a named MLP call (see net_bytes)."""
    calls = fine_calls(counts, counts.get("merged_off_pass", False)).get(name)
    if not calls:
        return None
    if name == "mlp_fwd(off)" and counts.get("merged_off_pass"):      # the detached on-tile half saves nothing
        i = NETS["off"][0]
        return (i * 4 + 16) * counts["n_on"] + net_bytes("off", "fwd", bf16) * counts["n_off"]
    if name == "mlp_fwd(rad)":          # off net: detached on-tile half saves nothing; emo net saves on the on-tiles
        i = NETS["off"][0]
        return ((i * 4 + 16) * counts["n_on"] + net_bytes("off", "fwd", bf16) * counts["n_off"]
                + net_bytes("emo", "fwd", bf16) * counts["n_on"])
    if name == "mlp_fwd(off|on-tiles)":
        return (NETS["off"][0] * 4 + 16) * counts["n_on"]
    return sum(net_bytes(n, op, bf16) * k for n, op, k in calls)


# engine call name -> HIP kernel (as rocprofv3 names it).  Only single-dispatch calls are roofline
# candidates; the batched wgrad call is several matrix + reduce kernels and is reported in all_mlp_kernels.
KERNEL_OF = {
    "mlp_fwd(off|on-tiles)": "mlp_fwd_kernel<0>", "mlp_fwd(off)": "mlp_fwd_kernel<0>",
    "mlp_fwd(emo)": "mlp_fwd_kernel<0>", "mlp_fwd(tone)": "mlp_fwd_kernel<1>",
    "mlp_dgrad(emo)": "mlp_dgrad_kernel<0>", "mlp_dgrad(off)": "mlp_dgrad_kernel<0>",
    "mlp_dgrad(tone)": "mlp_dgrad_kernel<1>",
    "mlp_fwd(rad)": "mlp_fwd_kernel<0>", "mlp_dgrad(rad)": "mlp_dgrad_kernel<0>",
    # fine stage, f32 engine: the step's eight weight-gradient jobs are ONE launch (+ a ~14 us slab reduction in the same call)
    "mlp_wgrad(all)": "mlp_wgrad_uni192_kernel",
}
# the same calls when the f32 engine runs them on the 16-bit matrix cores from split fp16 planes (round 4)
SPLIT_KERNELS = {"mlp_fwd(rad)": "mlp_fwd_split_kernel<0>", "mlp_dgrad(rad)": "mlp_dgrad_split_kernel<0>",
                 "mlp_wgrad(all)": "mlp_wgrad_uni192s_kernel"}


def best_threads(run, candidates):
    """Time ``run()`` (one iteration, seconds) under each torch thread count and keep the fastest: the default (all
    logical cores of the host) oversubscribes the small GEMMs of this path several-fold on a 128-256-thread box."""
    best = None
    for n in candidates:
        torch.set_num_threads(n)
        run()                                             # warm-up under this setting
        t = run()
        if best is None or t < best[1]:
            best = (n, t)
    torch.set_num_threads(best[0])
    return best[0]


def thread_candidates(with_all=True):
    ncpu = os.cpu_count() or 8
    return sorted({n for n in ((8, 16, 32, 64, ncpu) if with_all else (8, 16, 32, 64)) if n <= ncpu})


def cpu_baseline(model, scene, s_val, n_rays, iters):
    """The CPU port of the reference path (checker code) on a bounded sample."""
    from esr_nerf_amd.config import fine_cfg
    from oracle import fine_path as fp
    cfg = fine_cfg("cpu")
    c = fp.make_consts(cfg.app.model, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min, scene.mask_xyz_max,
                       scene.mask_alpha_init, scene.mask_density, scene.near, scene.num_voxels)
    P = fp.params_from_state_dict({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()})
    batch = {k: v[:n_rays].contiguous() for k, v in scene.batch.items()}
    probe = {k: v[:max(64, n_rays // 4)].contiguous() for k, v in scene.batch.items()}

    def one(b=batch):
        for v in P.values():
            v.grad = None
        t0 = time.perf_counter()
        res = fp.forward_training(P, c, b, s_val)
        loss, _ = fp.fine_loss(res, b["rgbs"])
        loss.backward()
        return time.perf_counter() - t0

    cand = thread_candidates()
    threads = best_threads(lambda: one(probe), cand)     # the thread count is picked on a quarter of the sample
    one()
    t = sum(one() for _ in range(iters)) / iters
    return dict(value=n_rays / t, unit="rays/s", cores=threads, kind="port",
                sample=f"{n_rays} of the {scene.n_rays} rays of the same scene (full grid), fwd+loss+bwd, "
                       f"{iters} iterations after warm-up, {t:.2f} s/iter, best of torch thread counts {cand} "
                       f"(host has {os.cpu_count()} logical cores), torch {torch.__version__} CPU ops + oracle/esr_oracle.c")


def lts_counts(eng):
    """The sizes ``lts_mlp_work`` prices one step with: taken right behind THAT step (survivor counts and the secondary pass
    change from step to step with the random surface points and directions)."""
    return dict(prim=dict(eng.prim.counts), pts_tiles=eng.pts.tiles_all, sec_m3=eng.sec.counts.get("m3", 0),
                wgrad_jobs=list(getattr(eng, "last_wgrad_jobs", [])))


def lts_mlp_work(breakdown, snaps, n_prof, bf16):
    """Algorithmic FLOPs, algorithmic HBM bytes and milliseconds per step of every MLP launch of an LTS / PDRA /
    fine-tune step, averaged over the instrumented steps (``snaps``: ``lts_counts`` behind each of them -- the times in
    ``breakdown`` are those steps' totals).  Call names: ``mlp_<op>(<net>)[<pass>]`` (samples per pass from the engine's
    survivor counts) and ``mlp_wgrad(all)`` = the batched weight gradients (jobs listed by the engine as (net[pass], tiles))."""
    fl = by = ms = 0.0
    for name, (n, t) in breakdown.items():
        if name.startswith("tone_wgrad[") or (name.startswith("mlp_") and "pack" not in name):
            ms += t / max(n_prof, 1)
    for c in snaps:
        pc = c["prim"]
        n_of = {"primary": {"off": pc["m3"], "emo": pc["n_on"], "tone": pc["m3"], "brdf": pc["m3"], "emit": pc["m3"]},
                "points": dict.fromkeys(NETS, c["pts_tiles"] * 32),
                "secondary": dict.fromkeys(NETS, c["sec_m3"]),
                "eps": dict.fromkeys(NETS, pc["m3"])}
        for name in breakdown:
            if name.startswith("tone_wgrad["):                       # f32 engine: tone weight gradients by recomputation
                k = n_of[name[len("tone_wgrad["):-1]]["tone"]
                fl += 2.0 * net_macs("tone", "wgrad") * k
                by += (NETS["tone"][0] * 4 + 16) * k
                continue
            if not name.startswith("mlp_") or "pack" in name:
                continue
            if name == "mlp_wgrad(all)":
                for job, tiles in c["wgrad_jobs"]:
                    net, pas = job.rstrip("]").split("[")
                    k = min(tiles * 32, n_of[pas][net])
                    fl += 2.0 * net_macs(net, "wgrad") * k
                    by += net_bytes(net, "wgrad", bf16) * k
                continue
            op, rest = name[4:].split("(", 1)
            net, pas = rest.split(")[")
            pas = pas.rstrip("]")
            fl += 2.0 * net_macs(net, op) * n_of[pas][net]
            by += net_bytes(net, op, bf16) * n_of[pas][net]
    return fl / max(len(snaps), 1), by / max(len(snaps), 1), ms


def cpu_baseline_lts(model, scene, s_val, n_rays, iters, stage, tr):
    """LTS / PDRA step of the CPU port on a proportionally scaled sample: n_rays primary rays and
    num_ltspts * n_rays / N surface points with the full 256 secondary rays each."""
    import numpy as np
    from esr_nerf_amd.config import lts_cfg
    from oracle import fine_path as fp
    from oracle import lts_path as lp
    cfg = lts_cfg("cpu")
    R = model.num_2ndrays
    Pn = max(1, int(round(model.num_ltspts * n_rays / scene.n_rays)))
    c = fp.make_consts(cfg.app.model, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min, scene.mask_xyz_max,
                       scene.mask_alpha_init, scene.mask_density, scene.near, scene.num_voxels)
    P = fp.params_from_state_dict({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()})
    batch = {k: v[:n_rays].contiguous() for k, v in scene.batch.items()}
    batch["uncert_masks"] = torch.arange(n_rays) % 3 == 0
    keep = {}
    fp.forward_training(fp.params_from_state_dict({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}),
                        c, batch, s_val, keep=keep)
    m3 = keep["counts"][3]
    g = torch.Generator().manual_seed(0)
    draws = lp.Draws(idx=torch.randperm(m3, generator=g)[:Pn], dirs=torch.randn(Pn, R + 1, 3, generator=g),
                     noise_normal=torch.randn(m3, 3, generator=g), noise_emit=torch.randn(m3, 3, generator=g))

    def one():
        for v in P.values():
            v.grad = None
        t0 = time.perf_counter()
        res = lp.forward_training(P, c, batch, s_val, draws, tr.normal_eps, tr.emit_eps, R, cfg.app.model.lts_near,
                                  pdra_mode=(stage == "pdra"))
        if stage == "pdra":
            loss, _ = lp.pdra_loss(res, batch["rgbs"], True, tr.weight_linear, tr.weight_lts, tr.weight_entropy_last,
                                   tr.weight_normal_smooth, tr.weight_emit_smooth, tr.weight_lts_l, tr.weight_lts_r,
                                   tr.weight_emit_supp)
        else:
            loss, _ = lp.lts_loss(res, batch["rgbs"], True, tr.weight_linear, tr.weight_lts, tr.weight_entropy_last,
                                  tr.weight_normal_smooth)
        loss.backward()
        return time.perf_counter() - t0

    # (the all-logical-cores setting is left out here: on the fine stage it is 4-5x slower than 16 threads on a 256-thread
    # host, and one LTS iteration under it takes minutes)
    cand = thread_candidates(with_all=False)
    threads = best_threads(one, cand)
    t = sum(one() for _ in range(iters)) / iters
    return dict(value=n_rays / t, unit="rays/s", cores=threads, kind="port",
                sample=f"{n_rays} of the {scene.n_rays} primary rays + {Pn} surface points x {R} secondary rays "
                       f"(same ratio as the full step), fwd+loss+bwd, {iters} iterations after warm-up, "
                       f"{t:.2f} s/iter, best of torch thread counts {cand} (host has {os.cpu_count()} logical cores), "
                       f"torch {torch.__version__} CPU ops + oracle/esr_oracle.c")


SCENE_OF = {"C2": "giftbox_w", "C3": "dtu scan97", "C4": "giftbox_w", "C5": "book_w"}


def pmc_tag(a, stage):
    """Which committed counter pass roofline.traffic / whole_step quote for this workload (profiles/pmc_traffic.json)."""
    side = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(side):
        return None
    with open(side) as f:
        wl = json.load(f).get("workloads", {}).get(f"{a.config}/{stage}/{a.dtype}/{float(a.s_val):g}")
    return {"source": "profiles/pmc_traffic.json", "tag": wl.get("_tag"), "passes": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
            "passes; FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE"} if wl else None


def pmc_traffic(a, stage, calls):
    """HBM bytes per launch of the named calls from the committed counter passes (profiles/pmc_traffic.json, one entry
    per workload: tools/pmc_summary.py), or None when no pass was taken on THIS workload.  ``__step__`` = all kernels
    of one step."""
    side = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(side):
        return None
    with open(side) as f:
        tj = json.load(f)
    wl = tj.get("workloads", {}).get(f"{a.config}/{stage}/{a.dtype}/{float(a.s_val):g}")
    if not wl:
        return None
    vals = [wl[c] for c in calls if c in wl]
    return sum(vals) / len(vals) if vals else None


# (name, bench.py arguments, environment of the child)
OTHER_WORKLOADS = (("C3_fine_bf16", ["--config", "C3", "--dtype", "bf16"], {}),
                   ("C4_lts_f32", ["--config", "C4"], {}),
                   ("C5_pdra_bf16", ["--config", "C5"], {}),
                   # the headline workload with every MLP product on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32), i.e. the
                   # engine of rounds 1-3: what the split-fp16 radiance kernels are measured against
                   ("C2_fine_f32_mfma_f32_only", [], {"ESR_SPLIT_FWD": "0"}))


def other_workloads(budget_s=170.0):
    """The other single-GPU BASELINE configs, each timed by a CHILD process of this script (30 steps after 10 warm-up
    steps -- with 10 + 6 the first timed steps still carried the warm-up's workspace growth and clock ramp: C3 bf16 read
    1.78 ms against 1.64 ms of the 50-step run -- no CPU baseline, no optimizer section) AFTER the headline has been measured: {name: {rays_per_s, ms_per_step,
    dtype, roofline fractions}}.  A child that fails or runs out of the time budget is reported as such, never guessed."""
    import subprocess
    out, t_start = {}, time.perf_counter()
    for name, args, env in OTHER_WORKLOADS:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20.0:
            out[name] = {"skipped": "time budget of the default run spent"}
            continue
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args, "--steps", "30", "--warmup", "10", "--no-cpu-baseline",
               "--no-optimizer", "--no-other"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, cwd=ROOT, env=dict(os.environ, **env))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"failed": (r.stderr or r.stdout)[-300:]}
                continue
            d = json.loads(line[-1])
            rl = d.get("roofline") or {}
            out[name] = {"rays_per_s": d["value"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "steps": d["steps"],
                         "warmup": d["warmup"], "workload": d["config"]["workload"],
                         "roofline": {k: rl.get(k) for k in ("bound", "kernel", "frac", "mfma_frac", "hbm_frac") if k in rl},
                         "whole_step": {k: v for k, v in (rl.get("whole_step") or {}).items()
                                        if k in ("mfma16_frac_issued", "mfma16_frac_algorithmic", "hbm_frac", "traffic_over_compulsory", "frac", "bound")},
                         "c_abi_launches_per_step": d.get("c_abi_launches_per_step")}
        except subprocess.TimeoutExpired:
            out[name] = {"failed": f"timeout after {left:.0f} s"}
    return out


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child, pass its output through (rank 0 prints
    the ONE JSON line last) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:                   # a free port for the rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    # stderr goes straight through; stdout is relayed line by line as it comes (a hung run shows how far it got), the JSON
    # line held back so that it is the LAST line of this process
    p = subprocess.Popen(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8")),
                         stdout=subprocess.PIPE, text=True, bufsize=1)
    js = None
    for l in p.stdout:
        l = l.rstrip("\n")
        if l.startswith("{"):
            if js is not None:
                print(js, flush=True)
            js = l
        else:
            print(l, flush=True)
    rc = p.wait()
    if js is not None:
        print(js, flush=True)                     # the ONE JSON line, last on stdout
    return rc if (rc != 0 or js is not None) else 1


def main():
    a = parse()
    bad_ranks = None
    headline = (a.config, a.stage, a.dtype, a.s_val, a.oblique, a.grid, a.gpus) == ("C2", None, None, None, False, None, 1)
    a.baseline_config = a.config                  # the label of the line (C5 runs on C4's scene objects)
    if a.config == "C5":
        a.config, a.stage, a.dtype = "C4", a.stage or "pdra", a.dtype or "bf16"
    a.dtype = a.dtype or "f32"
    if a.grid:
        if a.config != "C2":
            raise SystemExit("--grid applies to --config C2")
        a.config = f"C2g{a.grid}"
    stage = a.stage or ("lts" if a.config == "C4" else "fine")
    if a.s_val is None:
        a.s_val = 20.0 if stage == "fine" else 220.0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (this process has not touched
            # the GPU and must not be replaced by exec on this pool), relay rank 0's JSON line and the launcher's exit code
            raise SystemExit(self_launch(a.gpus))
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # Rehearsal of the N > 1 flow on a ONE-GPU box (ESR_BENCH_REHEARSAL=1): every rank uses device 0 and the ranks
    # talk over gloo, because RCCL refuses two ranks on one device.  Exercises the launch / barrier / max-over-ranks /
    # gradient-exchange code with real ranks; its numbers are NOT a measurement (two processes share one GPU).
    rehearsal = os.environ.get("ESR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    pg = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # RCCL prints a five-line banner (version, HIP / ROCm version, host name, library path) to STDOUT when its
            # communicator comes up; stdout is for rank 0's one JSON line, so the descriptor points at stderr meanwhile
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=torch.device(dev))     # nccl == RCCL on ROCm
                one = torch.ones(1, device=dev)
                dist.all_reduce(one)                 # (the communicator is created here at the latest)
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(keep, 1)
                os.close(keep)
        pg = dist.group.WORLD

    import numpy as np
    from esr_nerf_amd.config import fine_cfg, lts_cfg
    from esr_nerf_amd.esrnerf import ESRNeRF
    from esr_nerf_amd.synthetic import CONFIGS, init_slab_model, slab_scene
    from esr_nerf_amd.trainer import FineStep, LtsStep
    from esr_nerf_amd.voxurff import VoxurfF

    # identical parameters on every rank (seed 0), different rays per rank; weak scaling: the config's ray count on
    # every rank, strong scaling: the config's ray count split over the ranks
    per_rank = None
    if a.scaling == "strong" and world > 1:
        per_rank = CONFIGS[a.config]["n_rays"] // world
    scene = slab_scene(a.config, s_val=a.s_val, seed=rank, n_rays=per_rank, oblique=a.oblique)
    torch.manual_seed(0)
    np.random.seed(0)
    import contextlib
    import io
    cfg = fine_cfg(dev) if stage == "fine" else lts_cfg(dev)
    # (strong scaling of the LTS stages: the surface points are split over the ranks too -- LtsStep(split_points=True) --
    # so that the global secondary work equals the reference's single-process step)
    with contextlib.redirect_stdout(io.StringIO()):
        model = (VoxurfF if stage == "fine" else ESRNeRF)(
            cfg, scene.near, scene.far, scene.xyz_min, scene.xyz_max, scene.mask_xyz_min,
            scene.mask_xyz_max, scene.mask_alpha_init, scene.mask_density, scene.s_val, scene.num_voxels)
    init_slab_model(model, scene)
    model.mlp_dtype = a.dtype
    model.train()
    batch = {k: v.to(dev) for k, v in scene.batch.items()}
    n_rays = scene.n_rays
    if stage == "fine":
        step = FineStep(model, process_group=pg)
    elif stage == "finetune":
        # C5's second half (pdra.py:1047-1109, cfg/app/pdra.yaml:128-138): only emo_color / emo_rgbnet train, the target
        # is the edited emission + its light transport; batch = uncertain + certain rays with edit codes 0..4
        if pg is not None:
            raise SystemExit("--stage finetune is a single-GPU line (the reference fine-tunes one image at a time)")
        with torch.no_grad():
            model.brdf.grid.normal_(0.0, 0.1)
        for p_ in model.parameters():
            p_.requires_grad_(False)
        for p_ in list(model.emo_color.parameters()) + list(model.emo_rgbnet.parameters()):
            p_.requires_grad_(True)
        model.s_val = a.s_val
        model.train(True, finetune=True)
        from esr_nerf_amd.trainer import FinetuneStep
        g_ = torch.Generator().manual_seed(21)
        batch = dict(rays_o=batch["rays_o"], rays_d=batch["rays_d"], viewdirs=batch["viewdirs"],
                     em_modes=(torch.arange(n_rays, device=dev) % 5).long(),
                     em_intensities=(0.25 + 2.0 * torch.rand(n_rays, generator=g_)).to(dev),
                     em_colors=torch.rand(n_rays, 2, generator=g_).to(dev))
        step = FinetuneStep(model)              # (the direct driver; ESRNeRF.forward_finetune + autograd is the drop-in route)
    else:
        model.pdra_mode = stage == "pdra"
        with torch.no_grad():
            model.brdf.grid.normal_(0.0, 0.1)
        batch["uncert_masks"] = (torch.arange(n_rays, device=dev) % 3 == 0)
        step = LtsStep(model, cfg.app.trainer, stage=stage, process_group=pg, split_points=(a.scaling == "strong"))
    eng = model.engine
    if a.serial:
        eng.overlap_wgrad = False
        if hasattr(eng, "wgrad_early"):
            eng.wgrad_early, eng.scatter_streamed, eng.eps_stream = set(), set(), False

    # the trainers' do_tv lines: the same block, with the same weights and period, in all three training loops
    # (fine.py:383-400, lts.py:381-398, pdra.py:459-476; cfg/app/fine.yaml:73-83, lts.yaml:91-98, pdra.yaml:100-107);
    # the re-lighting fine-tune has none (pdra.py:1047-1109)
    tv_in_step = stage in ("fine", "lts", "pdra") and not a.no_tv
    TVS, W_TV, TV_EVERY = dict(sdf=0.1, smooth_grad=0.05), 0.01, 3

    def one(it=None):
        """One step; ``it``: iteration number of a timed / warm-up step -- on every TV_EVERY-th the trainer's do_tv
        lines (fine.py:383-400: smoothed-gradient TV value + gradient, in-place 6-neighbour TV gradient) run too."""
        if tv_in_step and it is not None and it % TV_EVERY == 0:
            return step.forward_loss_backward(batch, a.s_val, global_rays=n_rays * world if pg is not None else None,
                                              entropy_owner=(rank == world - 1),
                                              regularisers=dict(n_rays_global=n_rays * world, weight_tv_density=W_TV, tvs=TVS,
                                                                dense_mode=True))[:2]
        if stage == "finetune":                 # loss: 0.5 * mse(lin/pbr/emo, lin/pbr/emo_hat), pdra.py:1090-1093
            return step.forward_loss_backward(batch, a.s_val)
        # N > 1: this rank's rays are one shard of a global batch of n_rays * N rays
        return step.forward_loss_backward(batch, a.s_val, global_rays=n_rays * world if pg is not None else None,
                                          entropy_owner=(rank == world - 1))[:2]

    # Warm-up = W PLAIN steps, exactly the steps that are timed afterwards (round 5: until then up to three of them were the
    # instrumented ones below -- serialised kernels, events around every launch -- which left the device in another state
    # than the timed steps find it in: with the driver's short warm-up the first ~10 timed steps ran 5-20 % slower,
    # `--step-times`).  The per-kernel breakdown (HIP events around EVERY launch, kernels serialised: ~2 ms per step and a
    # read-back that idles the GPU) is taken in up to three EXTRA steps AFTER the timed region, like the optimizer and TV timings.
    # A generation-2 pass of Python's cyclic GC over this process (torch modules, thousands of tensors) takes
    # 50-110 ms -- twenty steps' worth of GPU idle when it lands inside a launch sequence (seen as a "112 ms
    # feat_bwd").  Collect now, park the survivors, and keep the collector off while kernels are being timed.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    n_prof = 0 if a.no_kernel_timing else 3
    for i_ in range(a.warmup):
        one(i_)
    # timed region: exactly K steps; the fine stage's three large launches carry one event pair each (on their stream) -- which
    # of them dominates is decided from the breakdown afterwards
    in_region = ["mlp_fwd(rad)", "mlp_dgrad(rad)", "mlp_wgrad(all)"] if (n_prof and stage == "fine") else None
    eng.enable_timing(in_region is not None, only=in_region)
    if pg is not None:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    calls0 = getattr(eng, "n_calls", 0)
    marks = []
    if a.step_times:
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[0].record()
    t0 = time.perf_counter()
    for i_ in range(a.steps):
        loss, _ = one(i_)
        if a.step_times:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    if pg is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if marks and rank == 0:       # (read back behind the clock: the line's value does not contain the read-backs; the per-step
        #  event records inside the loop remain -- config.step_times marks such a line)
        print("step times (ms, device):", [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(len(marks) - 1)], file=sys.stderr)
    calls_per_step = (getattr(eng, "n_calls", 0) - calls0) / max(a.steps, 1)
    if pg is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern = eng.timing_summary() if in_region else {}     # the bracketed launches: name -> (launches, total ms)
    eng.enable_timing(False)
    # the instrumented steps (outside W and K; every rank runs them: a step is collective when N > 1)
    breakdown, dominant, prof_counts = {}, None, []
    if n_prof:
        torch.cuda.synchronize()
        eng.enable_timing(True)
        overlap, eng.overlap_wgrad = eng.overlap_wgrad, False    # one kernel at a time while each is being timed
        for i_ in range(n_prof):
            one(i_)
            if stage != "fine":
                prof_counts.append(lts_counts(eng))
        breakdown = {k: (n, ms) for k, (n, ms) in eng.timing_summary().items()}
        eng.overlap_wgrad = overlap
        eng.enable_timing(False)
    if step is not None and hasattr(step, "close"):
        step.close()                                     # data parallel: the last step's deferred march-overflow flag (every rank raises)
    gc.enable()
    counts = dict(model.last_counts, merged_off_pass=(stage == "fine"))
    if n_prof:
        counts = dict(model.last_counts, merged_off_pass=(stage == "fine"))
        by_kernel = {}
        for call, kname in KERNEL_OF.items():
            if call in breakdown:
                by_kernel[kname] = by_kernel.get(kname, 0.0) + breakdown[call][1]
        split_fwd = bool(getattr(eng, "split_fwd", False)) and a.dtype == "f32" and stage == "fine"
        split_bwd = split_fwd and bool(getattr(eng, "split_bwd", False))
        split_wg = split_bwd and bool(getattr(eng, "split_wgrad", False))
        # the f32 engine's radiance launches run on the 16-bit matrix cores from split fp16 planes (csrc/mlp_split.hip,
        # wgrad_dma_body<SPLIT>): their kernel symbols
        for on, call, kname in ((split_fwd, "mlp_fwd(rad)", SPLIT_KERNELS["mlp_fwd(rad)"]),
                                (split_bwd, "mlp_dgrad(rad)", SPLIT_KERNELS["mlp_dgrad(rad)"]),
                                (split_wg, "mlp_wgrad(all)", SPLIT_KERNELS["mlp_wgrad(all)"])):
            if on and call in breakdown:
                by_kernel[KERNEL_OF[call]] = by_kernel.get(KERNEL_OF[call], 0.0) - breakdown[call][1]
                if by_kernel[KERNEL_OF[call]] <= 1e-9:
                    by_kernel.pop(KERNEL_OF[call])
                by_kernel[kname] = breakdown[call][1]
        if not (stage == "fine" and a.dtype == "f32"):
            by_kernel.pop("mlp_wgrad_uni192_kernel", None)             # (one launch only in the fine stage's f32 engine)
        elif not split_fwd:
            by_kernel.pop("mlp_wgrad_uni192_kernel", None)             # (rounds 1-3's choice of launch stays comparable)
        dominant = max(by_kernel, key=by_kernel.get) if by_kernel else None
        if dominant and a.dtype == "bf16":                     # the bf16 engine's kernel symbols (csrc/mlp_bf16.hip)
            dominant = dominant.replace("mlp_fwd_kernel", "mlp_fwd16s_kernel").replace("mlp_dgrad_kernel", "mlp_dgrad16s_kernel")
    split_of = {v: k for k, v in SPLIT_KERNELS.items()}
    if dominant in split_of:
        dom_calls = [split_of[dominant]]
    else:
        dom_calls = [c for c, k in KERNEL_OF.items() if dominant and k == dominant.replace("16s_kernel", "_kernel") and c in breakdown
                     and not (c == "mlp_fwd(rad)" and n_prof and split_fwd) and not (c == "mlp_dgrad(rad)" and n_prof and split_fwd and split_bwd)
                     and not (c == "mlp_wgrad(all)" and n_prof and split_wg)]
    # one launch of that kernel per step is bracketed (each event pair costs ~40-80 us of wall time)
    dom_calls = sorted(dom_calls, key=lambda c: -breakdown[c][1])[:1]
    if not dom_calls and not a.no_kernel_timing and stage == "fine":
        # too few warm-up steps for the breakdown: bracket the kernel that dominates every profile taken so far
        split_fwd = bool(getattr(eng, "split_fwd", False)) and a.dtype == "f32"
        split_bwd = split_fwd and bool(getattr(eng, "split_bwd", False))
        split_wg = split_bwd and bool(getattr(eng, "split_wgrad", False))
        dominant, dom_calls = ((SPLIT_KERNELS["mlp_fwd(rad)"], ["mlp_fwd(rad)"]) if split_fwd else
                               ("mlp_fwd_kernel<0>", ["mlp_fwd(rad)"]))
    if dominant and not all(c in kern for c in dom_calls):
        dominant, dom_calls = None, []                     # (not one of the bracketed launches: no in-region figure for it)

    # optimizer step, reported separately (outside the named path, never part of `value`)
    opt_ms = None
    if not a.no_optimizer and stage != "finetune":   # every rank: one() is a collective step when N > 1 (rank 0 reports)
        from esr_nerf_amd.optimizer import create_optimizer_or_freeze_model
        lrs = dict(off_color=0.1, off_rgbnet=0.003, emo_color=0.1, emo_rgbnet=0.003, sdf=0.005, tonemapper=0.003,
                   brdf=0.1, brdfnet=0.003, emitnet=0.003, envmap=0.003)
        opt = create_optimizer_or_freeze_model(model, **lrs)
        _, g_last = one()
        step.assign_grads(g_last)
        for _ in range(2):
            opt.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize()
        opt_ms = (time.perf_counter() - t1) / 5 * 1e3
        n_params = sum(p.numel() for g_ in opt.param_groups for p in g_["params"])
    # the per-step zero-fill of the flat gradient buffer (the reference's dense zeros_like(grid) allocations), timed alone:
    # it is inside every step (on the side stream, beside the march) and grows with the grid
    zero_ms = None
    if not a.no_optimizer and getattr(step, "_flat", None) is not None:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step._flat.zero_()
        torch.cuda.synchronize()
        zero_ms = (time.perf_counter() - t1) / 5 * 1e3
    # the trainer's every-third-iteration TV lines (fine.py:383-400), reported separately like the optimizer
    tv_ms = None
    if not a.no_optimizer and stage in ("fine", "lts", "pdra"):
        l_tv, g_tv = one()
        for _ in range(2):
            step.add_regularisers(l_tv, g_tv, n_rays * world, W_TV, TVS, True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step.add_regularisers(l_tv, g_tv, n_rays * world, W_TV, TVS, True)
        torch.cuda.synchronize()
        tv_ms = (time.perf_counter() - t1) / 5 * 1e3

    # fixed costs of the brick exchange (HIP events; EVERY rank: the phases contain collectives; with one rank they are
    # local): flags / union / list / pack / all-reduce / unpack -- what the driver's multi-GPU runs can be read against
    phases = None
    if pg is not None and step is not None and getattr(step, "_flat", None) is not None and not rehearsal:
        from esr_nerf_amd.grad_sync import GridGradSync
        prof = getattr(step, "_sync", None) or GridGradSync(pg)
        phases = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in prof.profile(step._flat[: step._n_grid]).items()}

    sweep = None
    if (pg is not None and (a.sync_sweep or (2 <= world <= 4 and not a.no_sync_sweep)) and stage != "finetune"
            and getattr(step, "sharded", None) is None):      # (a sharded optimizer forces the shard form: nothing to compare)
        sweep = {}
        keep_mode, keep_sync, keep_used = step._sync_mode, step._sync, getattr(step, "sync_mode_used", None)
        for mode in ("sparse", "dense"):
            step._sync_mode, step._sync = mode, None
            for _ in range(3):
                one()
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i_ in range(10):
                one(i_)
            torch.cuda.synchronize()
            dist.barrier()
            tt_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            sweep[mode] = float(tt_.item()) / 10 * 1e3
        step._sync_mode, step._sync, step.sync_mode_used = keep_mode, keep_sync, keep_used      # (what the line reports: the timed steps')
    if rank == 0:
        value = n_rays * world * a.steps / dt
        c = CONFIGS[a.config]
        samples = int(round(c["res"] * c["z"] * 2))
        out = {
            "metric": "training rays/sec at 4096 rays x 128 samples (fine stage)" if (a.config, stage, a.dtype, float(a.s_val), a.oblique) == ("C2", "fine", "f32", 20.0, False)
                      else f"training rays/sec, config {a.config if a.grid else a.baseline_config}, {stage} stage, {a.dtype} MLPs, s_val {a.s_val:g}" + (", OBLIQUE rays (not BASELINE's workload)" if a.oblique else "")
                           + (", PRODUCTION-SIZE GRID 256^3 (not BASELINE's workload)" if a.grid else ""),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)",
            "config": {
                # BASELINE.json's names: C2 / C4 giftbox_w, C3 dtu scan97, C5 book_w (pdra stage + the re-lighting fine-tune); the
                # data are the synthetic slab scene of that shape in every case (`data`)
                "workload": f"{a.baseline_config}: {SCENE_OF.get(a.baseline_config, 'giftbox_w')} {stage} stage on the slab scene, {n_rays} rays x {samples} "
                            f"samples per GPU, grid {'x'.join(str(int(v)) for v in model.world_size.tolist())}, "
                            f"s_val={a.s_val:g}, forward + trainer loss + backward"
                            + (" + the TV lines on every third step" if tv_in_step else "") + " (no optimizer step)"
                            + ("" if stage == "fine" else f"; + {getattr(step, 'ltspts', model.num_ltspts)} surface points x {model.num_2ndrays} "
                               f"secondary rays per GPU ({round(sum(c['sec_m3'] for c in prof_counts) / len(prof_counts)) if prof_counts else eng.sec.counts.get('m3')} "
                               f"surviving secondary samples, mean of the instrumented steps: the draws are random)")
                            + ("; fine-tune target: edited emission + its light transport, only emo_color / emo_rgbnet "
                               "train" if stage == "finetune" else ""),
                "rays_per_gpu": n_rays, "samples_per_ray": samples if not a.grid else 128, "surviving_samples": counts.get("m3"),
                "parallelism": f"dp{world}",
                **({"step_times": True} if a.step_times else {}),      # (events recorded inside the timed loop: a diagnostic line)
            },
            "loss": float(loss),
            # launches of this library's kernels per step (C-ABI calls; torch's own fills / copies / gathers come on top:
            # tools/dispatches.sh counts every device dispatch from a rocprofv3 trace)
            "c_abi_launches_per_step": calls_per_step,
            # steps that the split-fp16 kernels' range flag sent to the f32 MFMA kernels (fine_engine.py: same-step fallback);
            # a timed step among them would have run twice
            "split_fallback_steps": int(getattr(eng, "split_fallback_steps", 0)),
        }
        if opt_ms is not None:
            out["optimizer_step"] = {"ms": opt_ms, "parameters": n_params, "kernel": "esr_adam_step (fused Adam, 28 B/param)",
                                     "hbm_gbs": n_params * 28 / (opt_ms * 1e-3) / 1e9,
                                     "note": "reported separately, not part of value / ms_per_step"}
        if zero_ms is not None:
            out["grad_zero_fill"] = {"ms": zero_ms, "mb": step._flat.numel() * 4 / 1e6,
                                     "note": "one memset of the flat gradient buffer per step; inside value / ms_per_step"}
        if tv_ms is not None:
            out["tv_terms"] = {"ms": tv_ms, "every": TV_EVERY, "kernels": "esr_smooth_grad_tv_fwd/bwd + esr_tv_add_grad",
                               "in_timed_steps": tv_in_step,
                               "note": "do_tv lines of the trainer (fine.py:383-400, lts.py:381-398, pdra.py:459-476): run on every third TIMED step (part of "
                                       "value / ms_per_step) and timed alone here" if tv_in_step else
                                       "do_tv lines of the trainer (fine.py:383-400), timed alone; NOT in the timed steps (--no-tv)"}
        if dominant:
            launches = sum(kern[c][0] for c in dom_calls if c in kern)
            ms = sum(kern[c][1] for c in dom_calls if c in kern)
            flops_total = sum(algorithmic_flops(c, counts) * kern[c][0] for c in dom_calls if c in kern)
            ach = flops_total / (ms * 1e-3) / 1e12
            bytes_total = sum((algorithmic_bytes(c, counts, a.dtype == "bf16") or 0) * kern[c][0] for c in dom_calls if c in kern)
            traffic = pmc_traffic(a, stage, dom_calls)
            # matrix-pipe utilisation and effective clock of the same kernel from the committed counter pass
            # (tools/profile_mfma.sh -> profiles/*_mfma_util.csv; C2 fp32 only -- counters are per workload)
            pmc_mfma = None
            import csv
            import glob
            utils = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma_util.csv")))
            if utils and (a.config, stage, a.dtype, float(a.s_val)) == ("C2", "fine", "f32", 20.0):
                for r in csv.DictReader(open(utils[-1])):
                    if r["kernel"] == dominant:
                        pmc_mfma = {"source": os.path.relpath(utils[-1], ROOT), "mfma_busy_frac": float(r["mfma_busy_frac"]),
                                    "clock_ghz": float(r["clock_ghz"]),
                                    "issued_gflop_per_launch": float(r["mfma_issued_gflop_per_launch"])}
            total_ms = sum(ms_ for _, ms_ in breakdown.values()) or 1.0
            out["roofline"] = {
                "bound": "mfma", "kernel": dominant, "calls": dom_calls, "achieved": ach,
                "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE, "
                                                    "profiles/pmc_traffic.json, same workload)",
                "avg_launch_ms": ms / launches, "launches_timed": launches,
                "algorithmic_gflop_per_launch": flops_total / launches / 1e9,
                "share_of_kernel_time": sum(breakdown[c][1] for c in dom_calls if c in breakdown) / total_ms,
                "pmc": pmc_mfma,
            }
            if dominant in SPLIT_KERNELS.values():
                # products as split-fp16 triples on the 16-bit matrix cores: three ISSUED 16-bit MFMA FLOPs per algorithmic
                # FLOP, priced against THAT pipe's dense peak, and the algorithmic bytes against HBM.  Neither roof
                # reached to 0.6 = bound by the wave's in-order instruction issue (tools/ubench/split_stamps.hip,
                # issue_cost.hip): labelled "latency", `achieved` / `peak` / `frac` then quote the NEARER roof.
                gbs = bytes_total / (ms * 1e-3) / 1e9
                issued = 3.0 * ach
                hf, mf_ = gbs / HBM_PEAK_GBS, issued / MFMA_F16_PEAK_TF
                near_hbm = hf >= mf_
                out["roofline"].update({
                    "bound": "hbm" if hf >= 0.6 else "mfma" if mf_ >= 0.6 else "latency",
                    "achieved": gbs if near_hbm else issued, "peak": HBM_PEAK_GBS if near_hbm else MFMA_F16_PEAK_TF,
                    "unit": "GB/s" if near_hbm else "TFLOP/s", "frac": hf if near_hbm else mf_,
                    "hbm_frac": hf, "mfma_frac": mf_, "hbm_gbs_algorithmic": gbs,
                    "algorithmic_mb_per_launch": bytes_total / launches / 1e6,
                    "mfma16_tflops_issued": issued, "issued_16bit_gflop_per_launch": 3.0 * flops_total / launches / 1e9,
                    "fp32_equivalent_tflops": ach,
                    "pipe": "v_mfma_f32_32x32x16_f16 on two fp16 planes per fp32 operand (x = x1 + x2), three products per "
                            "algorithmic product, fp32 accumulation and fp32 results; error vs float64 as the f32 MFMA "
                            "kernels' (tests/test_gpu_split.py)",
                    "pmc": dict(pmc_tag(a, stage) or {}, **({"mfma_util": pmc_mfma} if pmc_mfma else {})) or None})
            elif a.dtype == "bf16":                # bf16 operands: the activation traffic, not the MFMA pipe, binds
                gbs = bytes_total / (ms * 1e-3) / 1e9
                hf, mf_ = gbs / HBM_PEAK_GBS, ach / MFMA_BF16_PEAK_TF
                # neither roof reached to 0.6: the kernel is bound by latency (barriers, epilogues), not by a roof -- say so;
                # `achieved` / `peak` / `frac` then quote the NEARER roof (HBM) and both fractions are printed
                out["roofline"].update({"bound": "hbm" if hf >= 0.6 else "mfma" if mf_ >= 0.6 else "latency",
                                        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hf,
                                        "hbm_frac": hf, "mfma_frac": mf_,
                                        "algorithmic_mb_per_launch": bytes_total / launches / 1e6,
                                        "mfma_tflops": ach, "mfma_frac_of_bf16_peak": mf_})
            # the whole MLP engine (all 10 calls per step), from the instrumented steps behind the timed region
            mlp = {k: v for k, v in breakdown.items() if algorithmic_flops(k, counts)
                   and not (split_fwd and k == "mlp_fwd(rad)") and not (split_fwd and split_bwd and k == "mlp_dgrad(rad)")
                   and not (split_wg and k == "mlp_wgrad(all)")
                   and not (split_fwd and k == "mlp_fwd(tone)" and 1 in getattr(eng, "split_kinds", ()))
                   and not (split_bwd and k == "mlp_dgrad(tone)" and 1 in getattr(eng, "split_kinds_bwd", ()))
                   and not (split_wg and k == "tone_wgrad" and getattr(eng, "split_tone_wgrad", False))}
            # (f32-pipe launches only: the split forward / input gradients are priced apart)
            peak = MFMA_BF16_PEAK_TF if a.dtype == "bf16" else MFMA_F32_PEAK_TF
            if mlp:
                mf = sum(algorithmic_flops(k, counts) * v[0] for k, v in mlp.items())
                mt = sum(v[1] for v in mlp.values()) * 1e-3
                out["roofline"]["all_mlp_kernels"] = {"achieved": mf / mt / 1e12, "frac": mf / mt / 1e12 / peak, "peak": peak,
                                                      "share_of_kernel_time": mt * 1e3 / total_ms,
                                                      "launches": sorted(mlp)}
                if a.dtype == "bf16":
                    mb = sum((algorithmic_bytes(k, counts, True) or 0) * v[0] for k, v in mlp.items())
                    out["roofline"]["all_mlp_kernels"].update(hbm_gbs=mb / mt / 1e9, hbm_frac=mb / mt / 1e9 / HBM_PEAK_GBS)
            elif split_fwd:
                out["roofline"]["all_mlp_kernels"] = {"launches": [], "note": "no launch of this step multiplies on the f32 MFMA pipe: every "
                                                      "net runs on the split-fp16 kernels (roofline.split_forward / split_dgrad / "
                                                      "split_wgrad price the three largest)"}
            if mlp or split_fwd:
                # the whole step against the path's roof: algorithmic FLOPs of the step / wall time
                # every net pass counted at the forward's MACs x 3 (fwd, dgrad, wgrad); an on-ray sample runs
                # the emo net x3, the off net x1 (detached) and the tone mapper x3, an off-ray sample the off net x3
                # and the tone mapper x3: 766,464 / 585,216 FLOP per sample = 86.5 MFLOP per ray at C2
                step_fl = (2 * (4 * RAD_MAC + 3 * TONE_MAC)) * counts["n_on"] + (2 * (3 * RAD_MAC + 3 * TONE_MAC)) * counts["n_off"]
                # exact MACs: the input-gradient pass multiplies the first layer only towards the grid-fed input rows
                # (43 of 85; 33 of 33 for the tone mapper) -- SURVEY's x3 convention over-counts it
                exact = sum(2 * net_macs(n_, op_) * k_ for call_ in ("mlp_fwd(rad)", "mlp_fwd(tone)", "mlp_dgrad(rad)",
                                                                       "mlp_dgrad(tone)", "mlp_wgrad(all)", "tone_wgrad")
                            for n_, op_, k_ in fine_calls(counts, True)[call_])
                if split_fwd and "mlp_fwd(rad)" in breakdown:
                    n_, t_ = breakdown["mlp_fwd(rad)"]
                    fl_, by_ = algorithmic_flops("mlp_fwd(rad)", counts), algorithmic_bytes("mlp_fwd(rad)", counts, False)
                    ms_ = t_ / n_
                    out["roofline"]["split_forward"] = {
                        "kernel": "mlp_fwd_split_kernel<0>", "avg_launch_ms": ms_,
                        "what": "the step's three radiance forward passes; every product as 3 fp16 MFMAs on two fp16 planes per "
                                "operand (x = x1 + x2), fp32 accumulation and fp32 results (csrc/mlp_split.hip)",
                        "algorithmic_gflop": fl_ / 1e9, "issued_16bit_gflop": 3 * fl_ / 1e9,
                        "mfma16_tflops_issued": 3 * fl_ / (ms_ * 1e-3) / 1e12,
                        "mfma16_frac": 3 * fl_ / (ms_ * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                        "hbm_gbs_algorithmic": by_ / (ms_ * 1e-3) / 1e9, "hbm_frac": by_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "fp32_equivalent_tflops": fl_ / (ms_ * 1e-3) / 1e12,
                        "note": "not priced against the f32 matrix peak: it does not run on that pipe"}
                if split_fwd and split_bwd and "mlp_dgrad(rad)" in breakdown:
                    n_, t_ = breakdown["mlp_dgrad(rad)"]
                    fl_, by_ = algorithmic_flops("mlp_dgrad(rad)", counts), algorithmic_bytes("mlp_dgrad(rad)", counts, False)
                    ms_ = t_ / n_
                    out["roofline"]["split_dgrad"] = {
                        "kernel": "mlp_dgrad_split_kernel<0>", "avg_launch_ms": ms_,
                        "what": "both radiance nets' input-gradient chains; products as 3 fp16 MFMAs on split planes, each tile "
                                "scaled by a power of two from its largest |dz|, fp32 dZ / dX (csrc/mlp_split.hip)",
                        "algorithmic_gflop": fl_ / 1e9, "issued_16bit_gflop": 3 * fl_ / 1e9,
                        "mfma16_frac": 3 * fl_ / (ms_ * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                        "hbm_gbs_algorithmic": by_ / (ms_ * 1e-3) / 1e9, "hbm_frac": by_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "fp32_equivalent_tflops": fl_ / (ms_ * 1e-3) / 1e12,
                        "note": "not priced against the f32 matrix peak: it does not run on that pipe"}
                if split_wg and "mlp_wgrad(all)" in breakdown:
                    n_, t_ = breakdown["mlp_wgrad(all)"]
                    fl_, by_ = algorithmic_flops("mlp_wgrad(all)", counts), algorithmic_bytes("mlp_wgrad(all)", counts, False)
                    ms_ = t_ / n_
                    out["roofline"]["split_wgrad"] = {
                        "kernel": "mlp_wgrad_uni192s_kernel (+ wgrad_reduce_kernel in the same call)", "avg_launch_ms": ms_,
                        "what": "every weight-gradient job of the two radiance nets in one launch; fp32 H / dZ / X tiles staged by "
                                "LDS-DMA, cut into two fp16 planes between LDS and the operand registers, three 16-bit MFMAs per "
                                "product block, fp32 accumulation (csrc/mlp.hip: wgrad_dma_body<SPLIT>)",
                        "bound": "hbm", "algorithmic_gflop": fl_ / 1e9, "algorithmic_mb": by_ / 1e6,
                        "hbm_gbs_algorithmic": by_ / (ms_ * 1e-3) / 1e9, "hbm_frac": by_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "mfma16_frac": 3 * fl_ / (ms_ * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                        "fp32_equivalent_tflops": fl_ / (ms_ * 1e-3) / 1e12}
                ws = out["roofline"]["whole_step"] = {"algorithmic_gflop": step_fl / 1e9, "exact_mac_gflop": exact / 1e9}
                if split_fwd:
                    # the step against the roofs of the pipes it runs on: issued 16-bit MFMA FLOPs (3 per algorithmic FLOP of
                    # the exact MAC count) over the fp16 matrix peak, and the step's HBM bytes (PMC) over its wall time.  (Until
                    # round 6 the line also carried the algorithmic rate over the F32 matrix peak -- 1.05-1.12, a pipe the
                    # step does not run on: not a roofline.)
                    ws["mfma16_frac_issued"] = 3 * exact / (dt / a.steps) / 1e12 / MFMA_F16_PEAK_TF
                    ws["mfma16_frac_algorithmic"] = exact / (dt / a.steps) / 1e12 / MFMA_F16_PEAK_TF
                else:
                    ws.update(achieved=step_fl / (dt / a.steps) / 1e12, peak=peak, frac=step_fl / (dt / a.steps) / 1e12 / peak,
                              frac_exact_mac=exact / (dt / a.steps) / 1e12 / peak)
                step_traffic = pmc_traffic(a, stage, ["__step__"])
                if step_traffic:
                    # compulsory bytes per iteration = rays, grids touched once, activations written once and read
                    # once by the backward: 1.31 KB per surviving sample + 80 B per ray = 0.69 GB at C2
                    compulsory = 1310.0 * counts["m3"] + 80.0 * n_rays
                    ws["hbm_frac"] = step_traffic / (dt / a.steps) / 1e9 / HBM_PEAK_GBS
                    ws["traffic_over_compulsory"] = step_traffic / compulsory
                    ws["hbm_bytes_pmc"] = step_traffic
                    ws["compulsory_bytes_survey_8d"] = compulsory
                    ws["pmc"] = pmc_tag(a, stage)
                    ws["bound"] = ("hbm" if ws["hbm_frac"] >= 0.6 else
                                   "mfma" if ws.get("mfma16_frac_issued", ws.get("frac", 0.0)) >= 0.6 else "latency")
                out["kernel_ms_per_step_instrumented"] = {k: round(v[1] / v[0], 4) for k, v in
                                                    sorted(breakdown.items(), key=lambda kv: -kv[1][1])}
        if breakdown and "kernel_ms_per_step_instrumented" not in out:
            out["kernel_ms_per_step_instrumented"] = {k: round(v[1] / max(n_prof, 1), 4) for k, v in
                                                sorted(breakdown.items(), key=lambda kv: -kv[1][1])}
        if breakdown and stage != "fine":
            bf = a.dtype == "bf16"
            fl, by, ms_mlp = lts_mlp_work(breakdown, prof_counts, n_prof, bf)
            total_ms = sum(ms_ for _, ms_ in breakdown.values()) / max(n_prof, 1)
            tf, gbs = fl / (ms_mlp * 1e-3) / 1e12, by / (ms_mlp * 1e-3) / 1e9
            peak = MFMA_BF16_PEAK_TF if bf else MFMA_F32_PEAK_TF
            rl = {"kernel": "all mlp_fwd/dgrad/wgrad launches of the step (HIP events, instrumented steps behind the timed region: every launch alone on "
                            "one stream; the timed steps run the backward on three streams)",
                  "traffic": pmc_traffic(a, stage, ["__step__"]), "mlp_ms_per_step": ms_mlp,
                  "share_of_kernel_time": ms_mlp / total_ms, "kernel_ms_per_step": total_ms,
                  "mfma": {"achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak},
                  "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "note": "algorithmic bytes of the MLP launches (inputs + saved activations / gradients)"}}
            # each line is priced against the roofs of the pipe its products RUN on: bf16 operands -> the bf16 matrix peak; f32
            # operands on the split-fp16 kernels -> three ISSUED 16-bit MFMA FLOPs per algorithmic FLOP against the fp16 matrix
            # peak; f32 operands on the f32 MFMA kernels (ESR_SPLIT_FWD=0) -> the f32 matrix peak.  Beside it the algorithmic
            # bytes against HBM.  Neither fraction at 0.6 = bound by launch / issue latency: labelled so, the NEARER roof quoted.
            split_lts = (not bf) and bool(getattr(eng, "split_fwd", False))
            if split_lts:
                rl["mfma"] = {"achieved": 3.0 * tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": 3.0 * tf / MFMA_F16_PEAK_TF,
                              "note": "issued 16-bit MFMA FLOPs (3 per algorithmic FLOP: split fp16 planes, fp32 results)",
                              "fp32_equivalent_tflops": tf}
            hf, mf_ = rl["hbm"]["frac"], rl["mfma"]["frac"]
            pick = rl["hbm"] if hf >= mf_ else rl["mfma"]
            rl.update(bound="hbm" if hf >= 0.6 else "mfma" if mf_ >= 0.6 else "latency",
                      achieved=pick["achieved"], peak=pick["peak"], unit=pick["unit"], frac=pick["frac"],
                      hbm_frac=hf, mfma_frac=mf_)
            out["roofline"] = rl
        if pg is not None:
            # how many ranks the collective library actually saw, and which exchange ran (trainer._grid_sync)
            out["dist"] = {"backend": dist.get_backend(pg), "rccl_ranks": dist.get_world_size(pg),
                           "grad_sync": getattr(step, "sync_mode_used", None),
                           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
            if out["dist"]["rccl_ranks"] != a.gpus:
                bad_ranks = f"--gpus {a.gpus} but the process group has {out['dist']['rccl_ranks']} ranks: not a {a.gpus}-GPU measurement"
        sync = getattr(step, "_sync", None)
        if phases is not None:
            out["grad_exchange_phases_ms"] = phases
        if sweep is not None:
            out["grad_sync_sweep_ms"] = dict(sweep, note="ms per step (10 steps after 3 warm-up steps) under each exchange form, same ranks")
        if sync is not None:
            # data-parallel exchange of the dense-grid gradients (esr_nerf_amd/grad_sync.py), last step of rank 0
            out["grad_exchange"] = dict(sync.last, brick_bytes=sync.brick * 4,
                                        dense_grid_mb=round(step._n_grid * 4 / 1e6, 1),
                                        sent_mb=round(sync.last["sent"] * sync.brick * 4 / 1e6, 1))
        elif pg is not None:
            out["grad_exchange"] = dict(mode="dense-async", dense_grid_mb=round(step._n_grid * 4 / 1e6, 1),
                                        sent_mb=round(step._n_grid * 4 / 1e6, 1))
        if world == 1 and not a.no_cpu_baseline:
            if stage == "fine":
                out["cpu_baseline"] = cpu_baseline(model, scene, a.s_val, min(a.cpu_rays or n_rays, n_rays), a.cpu_iters)
            elif stage != "finetune":
                out["cpu_baseline"] = cpu_baseline_lts(model, scene, a.s_val, min(a.cpu_rays or 1024, n_rays), a.cpu_iters,
                                                       stage, cfg.app.trainer)
        # (only the plain default run: profiling / A-B invocations of the headline workload pass one of the --no-* switches)
        if headline and world == 1 and not (a.no_other or a.force_dist or a.no_cpu_baseline or a.no_optimizer
                                            or a.no_kernel_timing or a.no_tv):
            # the other BASELINE configs, driver-observed: measured AFTER the headline (its fields above are final)
            out["other_workloads"] = other_workloads()
        line = json.dumps(out)
    if pg is not None:
        dist.destroy_process_group()
    if bad_ranks:
        raise SystemExit(bad_ranks)
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)       # RCCL's version banner sits in the C stdio buffer until exit: push it out first
        print(line, flush=True)              # the ONE JSON line, last on stdout


if __name__ == "__main__":
    main()
