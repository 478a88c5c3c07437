"""BASELINE.json's bar for the bf16 configurations (C3, C5): "PSNR within 0.1 dB of reference" -- as a teacher-student
experiment (tests/teacher_student.py): a fixed teacher renders the targets, a student is TRAINED on them with the
trainer steps + fused Adam + the trainers' cosine schedule, once with f32 MLP operands (the reference's arithmetic) and
once with bf16 operands, same seeds and batches, and scored by PSNR (utils2/metric.py:91-92) on held-out rays through
the image-rendering entry points.  The score moves by > 10 dB over training, so a bf16 forward / input-gradient /
weight-gradient kernel that lost precision would show up as a student that learns less.

Measured on MI355X (tools/psnr_teacher_student.py): fine stage, 100 steps: |f32 - bf16| <= 0.061 dB on seven of eight
seeds, f32 reruns within 0.006 dB (300 steps: the differences grow to ~0.1 dB and the runs start to bifurcate, see the
first test); fine-tune half -0.003 / -0.004 dB (reruns identical); pdra stage: chaotic per run, resolved by paired statistics over
seeds (round 4: 64 seeds, bf16 - f32 = +0.042 dB with a 95 % confidence interval of -0.152 .. +0.235 dB).
"""
import numpy as np
import pytest

import teacher_student as ts

pytestmark = pytest.mark.gpu

BAR_DB = 0.1            # BASELINE.json


def test_fine_stage_bf16_student_matches_f32_student_within_0p1_db():
    """C3's bar.  100 steps of the trainer's loop (cosine decay to zero) take the held-out score from ~24.5 dB to the
    ~37 dB plateau of this objective; there two f32 runs of the same seeds agree to 0.006 dB and the bf16 student to
    0.06 dB on seven of eight seeds measured (+0.004 / -0.034 / -0.022 / +0.028 / +0.002 / +0.061 / +0.013 dB).
    The trainer's objective is BISTABLE on a synthetic teacher (its linear-colour term assumes a gamma-curve tone mapper,
    the teacher's is a random MLP: the ~37 dB plateau is the compromise, and now and then a run finds the way past it).
    Round 4: whether a seed bifurcates is MEASURED, not attributed -- a seed whose bf16 student lands more than 0.1 dB
    from its f32 twin is run a third time in f32 with the initial MLP weights jittered by 1e-3 relative: the scale of the
    arithmetic difference under test (one bf16 rounding is up to 2^-9 = 2e-3 relative, 1.1e-3 rms), applied ONCE, where
    the bf16 engine rounds every operand of every step.  Only if THAT f32 run also lands more than 0.1 dB from the plain
    f32 run is the seed set aside: its outcome then does not resolve 0.1 dB at this perturbation scale whatever the
    operand type.  Measured on seed 6 (tools: fine_experiment(jitter=)): f32 42.89 dB, with jitter 1e-5 / 1e-4 / 1e-3 /
    4e-3: 42.92 / 43.27 / 44.98 / 46.79 dB; bf16 37.36 dB, with jitter 1e-3 / 4e-3: 43.03 / 46.47 dB -- the outcome
    moves by dB under 1e-3-scale perturbations in BOTH arithmetics (and bf16 + jitter lands where plain f32 does).
    Every other seed must be inside the bar, and at most two seeds may be set aside."""
    steps, seeds, diffs, aside = 100, (0, 1, 2, 3, 4, 5, 6), {}, {}
    for seed in seeds:
        r32, _, spread = ts.fine_experiment("f32", steps=steps, seed=seed)
        r16, _, _ = ts.fine_experiment("bf16", steps=steps, seed=seed)
        print(f"fine seed {seed}: f32 {r32[0]:.2f} -> {r32[steps]:.3f} dB, bf16 {r16[0]:.2f} -> {r16[steps]:.3f} dB")
        assert spread > 0.15                                       # the teacher's image has content
        for r in (r32, r16):
            assert r[steps] > r[0] + 8.0, r                        # the student learns: the score is sensitive
        d = r32[steps] - r16[steps]
        if abs(d) >= BAR_DB:
            rj, _, _ = ts.fine_experiment("f32", steps=steps, seed=seed, jitter=1e-3)
            dj = r32[steps] - rj[steps]
            print(f"fine seed {seed}: f32 - bf16 = {d:+.3f} dB; f32 with 1e-3 weight jitter lands {dj:+.3f} dB from plain f32")
            assert abs(dj) >= BAR_DB, (f"seed {seed}: bf16 is {d:+.3f} dB from f32 while a one-time 1e-3 perturbation of the "
                                       f"initial weights moves the f32 outcome by only {dj:+.3f} dB -- a precision difference, not a bifurcation")
            aside[seed] = (d, dj)
        else:
            diffs[seed] = d
    print("fine: f32 - bf16 per seed", {k: round(v, 3) for k, v in diffs.items()}, "set aside (measured bistable):", aside)
    assert len(aside) <= 2 and len(diffs) >= 5
    assert all(abs(d) < BAR_DB for d in diffs.values())
    assert abs(float(np.median(list(diffs.values())))) < BAR_DB


def test_finetune_half_bf16_matches_f32_within_0p1_db():
    """C5's second half (re-lighting fine-tune, pdra.py:1047-1109): only emo_color / emo_rgbnet train; scored as the
    reference reports it (loss2psnr of the fine-tune MSE) on held-out rays with fixed draws by ONE scorer (the f32
    engine), plus the two students' ``ESRNeRF.forward_evaluate`` images against each other."""
    steps = 80
    for seed in (0, 1):
        r32, l32, img32 = ts.finetune_experiment("f32", steps=steps, seed=seed)
        r16, l16, img16 = ts.finetune_experiment("bf16", steps=steps, seed=seed)
        print(f"finetune seed {seed}: f32 {r32[0]:.3f} -> {r32[steps]:.3f} dB, bf16 {r16[0]:.3f} -> {r16[steps]:.3f} dB, "
              f"images {ts.psnr(img32, img16):.1f} dB apart")
        assert r32[steps] > r32[0] + 0.5 and r16[steps] > r16[0] + 0.5          # the objective moves
        assert abs(r32[steps] - r16[steps]) < BAR_DB
        assert ts.psnr(img32, img16) > 50.0


def test_pdra_stage_bf16_minus_f32_paired_statistics():
    """C5's first half.  The stage's training is chaotic at this scale (pdra_experiment's docstring): two f32 runs of the
    same seeds -- differing only in float-atomic ordering -- end 0.55 dB (one sigma) apart, so a per-run 0.1 dB bar cannot
    be asserted.  It is resolved STATISTICALLY (round 4), by paired differences over seeds:
      * tools/psnr_teacher_student.py --stage pdra --seeds-range 64 --noise-floor (profiles/r04_psnr_pdra_64seeds.json):
        bf16 - f32 = +0.042 dB, 95 % CI [-0.152, +0.235]; f32 rerun - f32 = -0.018 dB, CI [-0.155, +0.119];
        256 seeds (profiles/r04_psnr_pdra_256seeds.json): see DESIGN.md section 5.
      * round 5, 512 seeds on the final build (profiles/r05_psnr_pdra.json, asserted by tests/test_psnr_statistics.py on
        every CPU run): the 95 % CI of the mean paired difference lies inside [-0.1, +0.1] dB.
      * here, on every GPU test run, 12 seeds x (f32, f32 rerun, bf16), what twelve seeds CAN decide:
          - per run: every trained student ends above 28 dB (random students start between 21 and 37 dB and the trained
            ones end between 28.5 and 35 dB: the minimum over the 2 x 256 runs of round 4 is 28.53), and a student that
            starts below 26 dB gains more than 3 dB (minimum over those 2 x 112 runs: 3.60; a student that happens to
            start at 35 dB cannot gain, which is why "start + 4 dB" is not a per-run property of this objective);
          - the mean gain is > 3 dB and mean(bf16) > mean(f32) - 1.5 dB;
          - the mean paired difference bf16 - f32 is inside 0.1 dB plus the half width of its own 99.9 % interval (the
            interval meets the bar's [-0.1, +0.1]; with 12 seeds that half width is ~1 dB -- the 512-seed file is what
            shrinks it to under 0.1), and the interval contains 0 (no systematic loss);
          - the scatter of bf16 - f32 is no more than 2.5x the scatter of two f32 runs of the same seed (measured ratio
            over 64 seeds: 1.41 -- the operand rounding is a larger initial perturbation than an atomic's summation
            order, in a system that amplifies both to the same attractor-sized spread).
    A bf16 kernel that lost precision costs several dB (the score moves by ~5 dB over training) and fails all of them."""
    steps, seeds = 200, range(12)
    d16, d32, gains, f32s, b16s = [], [], [], [], []
    for seed in seeds:
        ra, _, _ = ts.pdra_experiment("f32", steps=steps, seed=seed)
        rb, _, _ = ts.pdra_experiment("f32", steps=steps, seed=seed)
        rh, _, _ = ts.pdra_experiment("bf16", steps=steps, seed=seed)
        print(f"pdra seed {seed}: f32 {ra[0]:.2f} -> {ra[steps]:.3f} / rerun {rb[steps]:.3f} dB, bf16 -> {rh[steps]:.3f} dB")
        for r in (ra, rb, rh):
            assert r[steps] > 28.0, r
            assert r[0] >= 26.0 or r[steps] > r[0] + 3.0, r
        gains.append(ra[steps] - ra[0])
        f32s.append(ra[steps]); b16s.append(rh[steps])
        d16.append(rh[steps] - ra[steps])
        d32.append(rb[steps] - ra[steps])
    assert float(np.mean(gains)) > 3.0, gains       # the score is sensitive: training moves it by ~5 dB on average
    assert float(np.mean(b16s)) > float(np.mean(f32s)) - 1.5
    # 99.9 % intervals: with the true mean difference at 0 a 95 % interval would fail one run in twenty by construction
    s16, s32 = ts.paired_stats(d16, conf=0.999), ts.paired_stats(d32, conf=0.999)
    print(f"pdra: bf16 - f32 mean {s16['mean']:+.3f} dB (99.9 % CI {s16['ci95'][0]:+.3f} .. {s16['ci95'][1]:+.3f}, sd {s16['sd']:.3f}); "
          f"f32 rerun - f32 mean {s32['mean']:+.3f} dB (CI {s32['ci95'][0]:+.3f} .. {s32['ci95'][1]:+.3f}, sd {s32['sd']:.3f})")
    assert abs(s16["mean"]) < BAR_DB + s16["ci95_half_width"], s16
    assert s16["ci95"][0] <= 0.0 <= s16["ci95"][1], s16
    assert s16["sd"] <= 2.5 * max(s32["sd"], 0.3), (s16, s32)


def test_fine_stage_split_fp16_engine_trains_like_the_f32_mfma_engine():
    """Round 4: the f32 engine's radiance kernels form their products from split fp16 planes on the 16-bit matrix cores
    (csrc/mlp_split.hip, mlp.hip: wgrad_dma_body<SPLIT>), fp32 results.  The same training experiment as above with that
    engine ("f32") and with every product on the f32 MFMA pipe ("f32mfma": ESR_SPLIT_FWD=0, rounds 1-3's arithmetic):
    measured over 48 seeds (profiles/r04_psnr_fine_split_vs_mfma_48seeds.json) the difference is -0.0004 dB with a standard
    deviation of 0.0023 dB, two runs of the SAME engine differ by 0.0014 dB (float-atomic order).  Asserted here on four
    seeds: every pair within 0.02 dB -- a fifth of the bar the bf16 engine is held to, ten times the measured scatter."""
    steps = 100
    for seed in (0, 1, 2, 3):
        rs, _, _ = ts.fine_experiment("f32", steps=steps, seed=seed)
        rm, _, _ = ts.fine_experiment("f32mfma", steps=steps, seed=seed)
        print(f"fine seed {seed}: split-fp16 engine {rs[steps]:.4f} dB, f32-MFMA engine {rm[steps]:.4f} dB")
        assert rs[steps] > rs[0] + 8.0 and rm[steps] > rm[0] + 8.0
        assert abs(rs[steps] - rm[steps]) < 0.02, (seed, rs[steps], rm[steps])


@pytest.mark.parametrize("stage,steps,seeds,name", [
    # the fine-stage slice runs in EVERY `-m gpu` run (~1 GPU-minute, round 6): the driver's own run regenerates part of the
    # committed statistics on the build under test; the other two need --runslow
    ("fine", 100, 96, "r05_psnr_fine.json"),
    pytest.param("finetune", 80, 12, "r05_psnr_finetune.json", marks=pytest.mark.slow),
    pytest.param("pdra", 200, 96, "r05_psnr_pdra.json", marks=pytest.mark.slow)])
def test_committed_psnr_statistics_regenerate(stage, steps, seeds, name):
    """The statistics tests/test_psnr_statistics.py asserts come from profiles/r05_psnr_*.json; this regenerates a slice of each
    on the current build (~1 / 0.5 / 4 minutes of GPU time) and checks that the slice is a sample of the same
    distribution: its per-seed scores for the committed seeds agree where the stage is reproducible (fine-tune: to 0.05 dB),
    and its mean paired difference lies within three standard errors of the committed mean."""
    import json
    import math
    import os
    ref = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)))
    assert ref["stage"] == stage and ref["steps"] == steps
    run = dict(fine=ts.fine_experiment, finetune=ts.finetune_experiment, pdra=ts.pdra_experiment)[stage]
    by_seed = {r["seed"]: r for r in ref["per_seed"]}
    diffs = []
    for seed in range(seeds):
        r32 = run("f32", steps=steps, seed=seed)[0]
        r16 = run("bf16", steps=steps, seed=seed)[0]
        diffs.append(r16[steps] - r32[steps])
        if stage == "finetune" and seed in by_seed:
            assert abs(r32[steps] - by_seed[seed]["f32"]) < 0.05 and abs(r16[steps] - by_seed[seed]["bf16"]) < 0.05
    st = ts.paired_stats(diffs)
    se = max(ref["bf16_minus_f32"]["sd"], st["sd"]) / math.sqrt(seeds)
    print(f"{stage}: regenerated mean {st['mean']:+.4f} dB over {seeds} seeds (committed {ref['bf16_minus_f32']['mean']:+.4f} over "
          f"{ref['bf16_minus_f32']['n']}), standard error {se:.4f}")
    assert abs(st["mean"] - ref["bf16_minus_f32"]["mean"]) < 3.0 * se + 1e-3
