"""BASELINE.json's bar for the bf16 configurations (C3, C5) -- "PSNR within 0.1 dB of reference" -- asserted on the
COMMITTED statistics of the round's final build (profiles/r05_psnr_*.json, written by tools/psnr_teacher_student.py on the GPU
box: teacher-student training experiments, tests/teacher_student.py; regenerate with tools/final_profiles.sh).

What the bar means for a training experiment whose single runs are chaotic (two f32 runs of the same seeds end 0.5 dB
apart in the PDRA stage): the MEAN paired difference bf16 - f32 over seeds, with its 95 % confidence interval, lies inside
[-0.1, +0.1] dB.  Round 4's GPU test only asserted that an interval contained 0 -- true of any unresolved experiment."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAR_DB = 0.1


def _load(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        pytest.fail(f"{name} is not committed under profiles/ (tools/final_profiles.sh writes it)")
    with open(p) as f:
        return json.load(f)


@pytest.mark.parametrize("name,min_seeds", [("r05_psnr_pdra.json", 256), ("r05_psnr_fine.json", 256), ("r05_psnr_finetune.json", 16)])
def test_bf16_minus_f32_psnr_is_resolved_inside_the_bar(name, min_seeds):
    d = _load(name)
    st = d["bf16_minus_f32"]
    assert d["other"] == "bf16" and st["n"] >= min_seeds and abs(st["conf"] - 0.95) < 1e-9
    lo, hi = st["ci95"]
    print(f"{name}: bf16 - f32 = {st['mean']:+.4f} dB, 95 % CI [{lo:+.4f}, {hi:+.4f}] over {st['n']} seeds (sd {st['sd']:.3f}); "
          f"f32 gains {d['mean_gain_f32']:.1f} dB over training")
    assert -BAR_DB <= lo and hi <= BAR_DB, (name, st)
    assert d["mean_gain_f32"] > 0.5                      # the score moves over training: it is sensitive to the arithmetic
    # the statistics are those of the per-seed scores in the file
    diffs = [r["bf16"] - r["f32"] for r in d["per_seed"]]
    assert len(diffs) == st["n"] and abs(sum(diffs) / len(diffs) - st["mean"]) < 1e-9


def test_split_fp16_engine_trains_like_the_f32_mfma_engine():
    """The f32 engine's arithmetic (split fp16 planes, fp32 results) against every product on the f32 MFMA pipe, same training
    experiments: fine stage resolved to a hundredth of the bar; the PDRA stage is chaotic per run (two runs of ONE engine end
    0.55 dB apart, one sigma), so the file is a noise-floor record: the interval contains 0 and the scatter is the rerun scatter."""
    d = _load("r05_psnr_fine_split_vs_mfma.json")
    st = d["bf16_minus_f32"]                              # (the tool's key; here: f32mfma - split, d["other"] says so)
    assert d["other"] == "f32mfma" and st["n"] >= 200
    assert -0.01 < st["ci95"][0] and st["ci95"][1] < 0.01 and st["sd"] < 0.05, st
    p = _load("r05_psnr_pdra_split_vs_mfma.json")
    sp = p["bf16_minus_f32"]
    assert p["other"] == "f32mfma" and sp["n"] >= 300
    assert sp["ci95"][0] <= 0.0 <= sp["ci95"][1] and sp["ci95_half_width"] < 0.1 and sp["sd"] < 0.9, sp


def test_statistics_tool_merges_parts_and_rebuilds_a_summary_from_a_cut_call(tmp_path):
    """tools/psnr_teacher_student.py --merge (several calls' seeds into one file, a seed counted once, statistics recomputed)
    and --from-log (a call cut at its time limit: the per-run lines it printed) -- how profiles/r05_psnr_*.json were put
    together from calls that each fit the GPU box's time limit."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "psnr_teacher_student.py")

    def part(seeds, path):
        rows = [dict(seed=s, f32=30.0 + 0.1 * s, bf16=30.0 + 0.1 * s + (0.05 if s % 2 else -0.03), f32_rerun=None, start=25.0)
                for s in seeds]
        with open(path, "w") as f:
            json.dump(dict(stage="pdra", steps=200, seeds=len(rows), other="bf16", per_seed=rows), f)
    a, b, out = tmp_path / "a.json", tmp_path / "b.json", tmp_path / "m.json"
    part(range(0, 6), a)
    part(range(4, 10), b)                                   # seeds 4, 5 in both: counted once
    r = subprocess.run([sys.executable, tool, "--merge", str(a), str(b), "--summary", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out))
    assert d["seeds"] == 10 and [p["seed"] for p in d["per_seed"]] == list(range(10)) and d["stage"] == "pdra" and d["steps"] == 200
    st = d["bf16_minus_f32"]
    assert st["n"] == 10 and abs(st["mean"] - 0.01) < 1e-12 and st["ci95"][0] < st["mean"] < st["ci95"][1]
    assert abs(d["mean_gain_f32"] - (5.0 + 0.45)) < 1e-9
    log = tmp_path / "run.log"
    with open(log, "w") as f:
        f.write("voxel_size 0.1\n")
        for s in range(3):
            for tag, v in (("f32", 31.0 + s), ("bf16", 31.25 + s)):
                f.write(json.dumps(dict(stage="fine", seed=s, dtype=tag, psnr={"0": 24.5, "50": 30.0, "100": v})) + "\n")
            f.write(f"seed {s}: final PSNR ...\n")
        f.write(json.dumps(dict(stage="fine", seed=3, dtype="f32", psnr={"0": 24.5, "100": 35.0})) + "\n")     # cut mid-seed
    out2 = tmp_path / "l.json"
    r = subprocess.run([sys.executable, tool, "--from-log", str(log), "--summary", str(out2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out2))
    assert d["stage"] == "fine" and d["steps"] == 100 and d["seeds"] == 3
    assert abs(d["bf16_minus_f32"]["mean"] - 0.25) < 1e-12 and all(p["start"] == 24.5 for p in d["per_seed"])
