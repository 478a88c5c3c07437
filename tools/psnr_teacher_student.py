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
ap.add_argument("--seed-start", type=int, default=0, help="with --seeds-range: seeds START .. N-1 (a long run in several calls)")
ap.add_argument("--from-log", default=None,
                help="no runs: rebuild --summary from the per-run lines of this tool's own output (a call that was cut at its time limit)")
ap.add_argument("--merge", nargs="+", default=None,
                help="no runs: merge the per-seed scores of these summary files (same stage / steps / other; a seed counted once) "
                     "into --summary, with the statistics recomputed")
ap.add_argument("--summary", default=None,
                help="write paired statistics (bf16 - f32 and, with --noise-floor, f32 rerun - f32: mean, sd, 95 %% CI of the "
                     "mean) and the per-seed scores to this JSON file")
a = ap.parse_args()
if a.seeds_range:
    a.seeds = list(range(a.seed_start, a.seeds_range))
per_seed = []


def summarise(per_seed):
    out = dict(stage=a.stage, steps=a.steps, seeds=len(per_seed), other=a.other,
               bf16_minus_f32=ts.paired_stats(r["bf16"] - r["f32"] for r in per_seed),
               mean_f32=sum(r["f32"] for r in per_seed) / len(per_seed), mean_bf16=sum(r["bf16"] for r in per_seed) / len(per_seed),
               mean_gain_f32=sum(r["f32"] - r["start"] for r in per_seed) / len(per_seed))
    if all(r.get("f32_rerun") is not None for r in per_seed):
        out["f32rerun_minus_f32"] = ts.paired_stats(r["f32_rerun"] - r["f32"] for r in per_seed)
    return out


def write_summary(per_seed, quiet=False):
    out = summarise(per_seed)
    if not quiet:
        print(json.dumps(out), flush=True)
    if a.summary:
        out["per_seed"] = per_seed
        with open(a.summary + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(a.summary + ".tmp", a.summary)


if a.from_log:
    # the per-run lines this tool prints ({"stage", "seed", "dtype", "psnr": {step: dB}}; scores rounded to 0.001 dB)
    runs = {}
    for line in open(a.from_log):
        if line.startswith("{") and '"dtype"' in line:
            r = json.loads(line)
            a.stage = r["stage"]
            runs.setdefault(r["seed"], {})[r["dtype"]] = {int(k): v for k, v in r["psnr"].items()}
    a.steps = max(max(v) for rr in runs.values() for v in rr.values())
    rows = [dict(seed=k, f32=rr["f32"][a.steps], bf16=rr["bf16"][a.steps],
                 f32_rerun=rr["f32b"][a.steps] if "f32b" in rr else None, start=rr["f32"][0])
            for k, rr in sorted(runs.items()) if "f32" in rr and "bf16" in rr]
    write_summary(rows)
    sys.exit(0)

if a.merge:
    seen = {}
    for path in a.merge:
        part = json.load(open(path))
        a.stage, a.steps, a.other = part["stage"], part["steps"], part.get("other", "bf16")
        for r in part["per_seed"]:
            seen.setdefault(r["seed"], r)
    write_summary([seen[k] for k in sorted(seen)])
    sys.exit(0)

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
        write_summary(per_seed, quiet=seed != a.seeds[-1])      # after every seed: a call cut short keeps what it measured
