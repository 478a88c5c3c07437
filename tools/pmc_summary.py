#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_workload.sh into the committed summaries:

    python tools/pmc_summary.py <tag> [--config C2 --stage fine --dtype f32 --s-val 20]

  profiles/<tag>_kernel_stats.csv     the --stats table (helper kernels below 0.05 % dropped)
  profiles/<tag>_pmc_fetch_size.csv   per-dispatch FETCH_SIZE of the LAST profiled step (our kernels only)
  profiles/<tag>_pmc_write_size.csv   same for WRITE_SIZE
  profiles/pmc_traffic.json           workloads/<config>/<stage>/<dtype>/<s_val>: HBM bytes per launch of every kernel of a
                                      step = FETCH_SIZE (KB) x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) +
                                      WRITE_SIZE (KB); "__step__" = all kernels of one step; engine call names for the
                                      single-dispatch MLP calls of the fine stage (what bench.py's roofline.traffic reads)

Dispatches of one kernel symbol are mapped to the engine's call names by their order inside a step."""
import argparse
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--config", default="C2")
ap.add_argument("--stage", default=None)
ap.add_argument("--dtype", default=None)
ap.add_argument("--s-val", type=float, default=None)
a = ap.parse_args()
if a.config == "C5":
    a.config, a.stage, a.dtype = "C4", a.stage or "pdra", a.dtype or "bf16"
a.dtype = a.dtype or "f32"
a.stage = a.stage or ("lts" if a.config == "C4" else "fine")
a.s_val = a.s_val if a.s_val is not None else (20.0 if a.stage == "fine" else 220.0)
tag = a.tag
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
# order of the single-dispatch MLP calls of one fine-stage step, per kernel symbol (f32 engine: the off net's detached
# and saved forward passes are ONE launch; bf16 engine: two)
# (round 3: the three radiance forward passes are one launch, esr_mlp_fwd_fine, and so are the two input-gradient passes)
# (round 4: the radiance launches of the f32 engine are the split-fp16 kernels)
ORDER_F32 = {"mlp_fwd_kernel<0>": ["mlp_fwd(rad)"], "mlp_dgrad_kernel<0>": ["mlp_dgrad(rad)"],
             "mlp_fwd_split_kernel<0>": ["mlp_fwd(rad)"], "mlp_dgrad_split_kernel<0>": ["mlp_dgrad(rad)"],
             "mlp_fwd_split_kernel<1>": ["mlp_fwd(tone)"], "mlp_dgrad_split_kernel<1>": ["mlp_dgrad(tone)"],
             "mlp_wgrad_uni192s_kernel": ["mlp_wgrad(all)"], "mlp_wgrad_uni192_kernel": ["mlp_wgrad(all)"],
             "mlp_fwd_kernel<1>": ["mlp_fwd(tone)"], "mlp_dgrad_kernel<1>": ["mlp_dgrad(tone)"]}
ORDER_BF16 = {"mlp_fwd16s_kernel<0>": ["mlp_fwd(rad)"], "mlp_fwd16s_kernel<1>": ["mlp_fwd(tone)"],
              "mlp_dgrad16s_kernel<0>": ["mlp_dgrad(rad)"], "mlp_dgrad16s_kernel<1>": ["mlp_dgrad(tone)"],
              "mlp_fwd16_kernel<1>": ["mlp_fwd(tone)"], "mlp_dgrad16_kernel<1>": ["mlp_dgrad(tone)"]}
ORDER = (ORDER_BF16 if a.dtype == "bf16" else ORDER_F32) if a.stage == "fine" else {}


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def symbol(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:60]


stats = one(f"{tag}_stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(stats)))
with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if float(r[4]) >= 0.05:
            w.writerow(r)

import shutil
byg = glob.glob(os.path.join(G, f"{tag}_stats", "**", "kernel_by_grid.csv"), recursive=True)
if byg:
    shutil.copy(byg[0], os.path.join(P, f"{tag}_kernel_by_grid.csv"))

per_call, per_kernel, step = {}, {}, {}
for what in ("fetch", "write"):
    src = one(f"{tag}_{what}/**/*counter_collection.csv")
    rd = list(csv.DictReader(open(src)))
    ours = [r for r in rd if "anonymous namespace" in r["Kernel_Name"]]
    # last step = the last complete run of dispatches starting at the final march_kernel<0> of the primary rays
    first = [i for i, r in enumerate(ours) if "march_kernel<0" in r["Kernel_Name"]]
    per_step = 2 if a.stage != "fine" else 1                       # LTS stages: primary + secondary march per step
    last = ours[first[-per_step]:]
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
            "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(P, f"{tag}_pmc_{what}_size.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep)
        w.writeheader()
        for r in last:
            w.writerow({k: r[k] for k in keep})
    seen = {}
    for r in last:
        k = symbol(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        step[what] = step.get(what, 0.0) + v
        per_kernel.setdefault(k, {}).setdefault(what, []).append(v)
        i = seen.get(k, 0)
        seen[k] = i + 1
        if k in ORDER and i < len(ORDER[k]):
            per_call.setdefault(ORDER[k][i], {})[what] = v

side = os.path.join(P, "pmc_traffic.json")
out = json.load(open(side)) if os.path.exists(side) else {}
if "workloads" not in out:
    out = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                    "tools/profile_workload.sh). FETCH_SIZE (KB) is doubled per the gfx950 correction of "
                    "MI355X_MICROARCH.md (HBM section); WRITE_SIZE (KB) is taken as is. bench.py reports these as "
                    "roofline.traffic for the matching workload only.", "workloads": {}}
hbm = lambda f, w: f * 1024 * 2 + w * 1024
key = f"{a.config}/{a.stage}/{a.dtype}/{a.s_val:g}"
wl = {"_tag": tag, "__step__": hbm(step["fetch"], step["write"]),
      "_kernels": {k: {"launches": len(v["fetch"]), "hbm_bytes_per_step": hbm(sum(v["fetch"]), sum(v.get("write", [0.0])))}
                   for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]["fetch"])) if "write" in v}}
for call, v in per_call.items():
    if "fetch" in v and "write" in v:
        wl[call] = hbm(v["fetch"], v["write"])
out["workloads"][key] = wl
json.dump(out, open(side, "w"), indent=1)
print(key, json.dumps({k: v for k, v in wl.items() if not k.startswith("_k")}, indent=1))
for k, v in list(wl["_kernels"].items())[:12]:
    print(f"  {k:45s} {v['launches']:3d} launches  {v['hbm_bytes_per_step'] / 1e6:9.1f} MB per step")
