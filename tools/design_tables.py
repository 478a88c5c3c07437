#!/usr/bin/env python3
"""Numbers for DESIGN.md section 6 from profiles/<tag>_bench_*.json (and the start-of-round set for the comparison column):
   python tools/design_tables.py [r04_z] [r04_a]"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_z"
old = sys.argv[2] if len(sys.argv) > 2 else "r04_a"


def load(t, name):
    p = os.path.join(ROOT, "profiles", f"{t}_bench_{name}.json")
    if not os.path.exists(p):
        return None
    return json.loads(open(p).read().strip().splitlines()[-1])


for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_bench_*.json"))):
    name = os.path.basename(f)[len(tag) + 7:-5]
    d, o = load(tag, name), load(old, name)
    r = d.get("roofline") or {}
    ws = (r.get("whole_step") or {}).get("mfma16_frac_issued") or (r.get("whole_step") or {}).get("frac")
    was = f"{o['ms_per_step']:.2f} ms, {o['value'] / 1e6:.2f} M" if o else "-"
    print(f"{name:24s} {d['ms_per_step']:.2f} ms  {d['value'] / 1e6:.2f} M rays/s | start {was} | {r.get('bound')} "
          f"{(r.get('kernel') or '')[:26]} frac {r.get('frac') or 0:.2f} hbm {r.get('hbm_frac') or 0:.2f} mfma {r.get('mfma_frac') or 0:.2f}"
          + (f" whole-step (issued 16-bit MFMA or own pipe) {ws:.2f}" if ws else ""))
    if name == "c2_f32":
        for k, v in (d.get("kernel_ms_per_step_instrumented") or d.get("kernel_ms_per_step_warmup", {})).items():
            print(f"      {k:18s} {v:.3f}")
        for k in ("split_forward", "split_dgrad", "split_wgrad"):
            b = r.get(k) or {}
            print(f"      {k}: {b.get('avg_launch_ms', 0):.3f} ms  hbm {b.get('hbm_frac', 0):.2f}  mfma16 {b.get('mfma16_frac', 0):.2f}  "
                  f"fp32-equivalent {b.get('fp32_equivalent_tflops', 0):.0f} TF")
        print("      timed-region launch", r.get("avg_launch_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              "optimizer", (d.get("optimizer_step") or {}).get("ms"), "tv", (d.get("tv_terms") or {}).get("ms"))
