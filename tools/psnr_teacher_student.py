#!/usr/bin/env python3
"""Held-out PSNR of a student trained with f32 and with bf16 MLP operands against a fixed teacher's image
(tests/teacher_student.py).  Run on the GPU box:
    python tools/psnr_teacher_student.py [--steps 300] [--seeds 0 1 2] [--stage fine|pdra]
Prints one line per (seed, dtype): PSNR at the evaluation steps; the difference f32 - bf16 at the end; and a second
f32 run per seed (same seeds, float-atomic ordering differs) as the noise floor of the comparison."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import teacher_student as ts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--seeds", type=int, nargs="+", default=[0])
ap.add_argument("--stage", default="fine", choices=["fine", "pdra", "finetune"])
ap.add_argument("--sdf-lr", type=float, default=None, help="fine stage: override the SDF grid learning rate")
ap.add_argument("--perturb", type=float, nargs=2, default=None, help="fine stage: student = teacher + (grid sigma, weight sigma)")
ap.add_argument("--weight-linear", type=float, default=None)
ap.add_argument("--lattice", type=int, nargs=3, default=None)
ap.add_argument("--noise-floor", action="store_true", help="also a second f32 run per seed")
ap.add_argument("--other", default="bf16", choices=["bf16", "f32mfma"],
                help="the arithmetic compared with the f32 engine: bf16 MLP operands (default), or f32mfma = the f32 engine with "
                     "every product on the f32 MFMA pipe (ESR_SPLIT_FWD=0) -- then 'f32' is the split-fp16 engine under test")
ap.add_argument("--seeds-range", type=int, default=None, help="seeds 0 .. N-1 (overrides --seeds)")
ap.add_argument("--summary", default=None,
                help="write paired statistics (bf16 - f32 and, with --noise-floor, f32 rerun - f32: mean, sd, 95 %% CI of the "
                     "mean) and the per-seed scores to this JSON file")
a = ap.parse_args()
if a.seeds_range:
    a.seeds = list(range(a.seeds_range))
per_seed = []
run = dict(fine=ts.fine_experiment, pdra=ts.pdra_experiment, finetune=ts.finetune_experiment)[a.stage]
ev = sorted({0, a.steps // 4, a.steps // 2, 3 * a.steps // 4, a.steps})
for seed in a.seeds:
    res = {}
    for tag, dt in (("f32", "f32"), ("bf16", a.other)) + ((("f32b", "f32"),) if a.noise_floor else ()):
        kw = dict(lrs=dict(ts.LRS_FINE, sdf=a.sdf_lr)) if (a.sdf_lr is not None and a.stage == 'fine') else {}
        if a.weight_linear is not None and a.stage == 'fine':
            kw['weight_linear'] = a.weight_linear
        if a.lattice is not None and a.stage == 'fine':
            kw['lattice'] = tuple(a.lattice)
        if a.perturb is not None and a.stage == 'fine':
            kw['perturb'] = tuple(a.perturb)
        scores, losses, spread = run(dt, steps=a.steps, seed=seed, eval_at=ev, **kw)
        res[tag] = scores
        if a.stage == "finetune":
            res[tag + "_img"], spread = spread, float(spread.std())
        print(json.dumps(dict(stage=a.stage, seed=seed, dtype=tag, psnr={k: round(v, 3) for k, v in scores.items()},
                              loss0=round(losses[0], 5), lossN=round(sum(losses[-10:]) / 10, 6), img_std=round(spread, 3))),
              flush=True)
    d = res["f32"][a.steps] - res["bf16"][a.steps]
    line = f"seed {seed}: final PSNR f32 {res['f32'][a.steps]:.3f}  bf16 {res['bf16'][a.steps]:.3f}  diff {d:+.3f} dB"
    if a.noise_floor:
        line += f"  | f32 rerun diff {res['f32'][a.steps] - res['f32b'][a.steps]:+.3f} dB"
    if a.stage == "finetune":
        line += f"  | images f32 vs bf16: {ts.psnr(res['f32_img'], res['bf16_img']):.1f} dB"
    print(line, flush=True)
    per_seed.append(dict(seed=seed, f32=res["f32"][a.steps], bf16=res["bf16"][a.steps],
                         f32_rerun=res["f32b"][a.steps] if a.noise_floor else None, start=res["f32"][0]))
if len(per_seed) > 1:
    out = dict(stage=a.stage, steps=a.steps, seeds=len(per_seed), other=a.other,
               bf16_minus_f32=ts.paired_stats(r["bf16"] - r["f32"] for r in per_seed),
               mean_f32=sum(r["f32"] for r in per_seed) / len(per_seed), mean_bf16=sum(r["bf16"] for r in per_seed) / len(per_seed),
               mean_gain_f32=sum(r["f32"] - r["start"] for r in per_seed) / len(per_seed))
    if a.noise_floor:
        out["f32rerun_minus_f32"] = ts.paired_stats(r["f32_rerun"] - r["f32"] for r in per_seed)
    print(json.dumps({k: v for k, v in out.items()}), flush=True)
    if a.summary:
        out["per_seed"] = per_seed
        with open(a.summary, "w") as f:
            json.dump(out, f, indent=1)
