#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_c2.sh into the committed summaries:

    python tools/pmc_summary.py r01_d      (reads gpurun_out/r01_d_{stats,fetch,write}/, writes profiles/)

  profiles/<tag>_kernel_stats.csv     the --stats table (hip/torch helper kernels below 0.05 % dropped)
  profiles/<tag>_pmc_fetch_size.csv   per-dispatch FETCH_SIZE of the LAST profiled step (our kernels only)
  profiles/<tag>_pmc_write_size.csv   same for WRITE_SIZE
  profiles/pmc_traffic.json           HBM bytes per launch of every MLP forward / dgrad call: FETCH_SIZE (KB) x 2
                                      (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB)

Dispatches of one kernel symbol are mapped to the engine's call names by their order inside a step
(forward: off|on-tiles, off, emo; backward: emo, off)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
ORDER = {"mlp_fwd_kernel<0>": ["mlp_fwd(off|on-tiles)", "mlp_fwd(off)", "mlp_fwd(emo)"],
         "mlp_dgrad_kernel<0>": ["mlp_dgrad(emo)", "mlp_dgrad(off)"],
         "mlp_fwd_kernel<1>": ["mlp_fwd(tone)"], "mlp_dgrad_kernel<1>": ["mlp_dgrad(tone)"],
         "feat_bwd_kernel": ["feat_bwd"], "feat_fwd_kernel": ["feat_fwd"]}


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def short(name):
    for k in ORDER:
        if k in name:
            return k
    return None


stats = one(f"{tag}_stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(stats)))
with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if float(r[4]) >= 0.05:
            w.writerow(r)

per_call = {}
for what in ("fetch", "write"):
    src = one(f"{tag}_{what}/**/*counter_collection.csv")
    rd = list(csv.DictReader(open(src)))
    ours = [r for r in rd if "anonymous namespace" in r["Kernel_Name"]]
    # last step = the last complete run of dispatches starting at the final march_kernel<0>
    starts = [i for i, r in enumerate(ours) if "march_kernel<0" in r["Kernel_Name"]]
    last = ours[starts[-1]:]
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
            "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(P, f"{tag}_pmc_{what}_size.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep)
        w.writeheader()
        for r in last:
            w.writerow({k: r[k] for k in keep})
    seen = {}
    for r in last:
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        i = seen.get(k, 0)
        seen[k] = i + 1
        if i < len(ORDER[k]):
            per_call.setdefault(ORDER[k][i], {})[what] = float(r["Counter_Value"])

out = {"_workload": {"config": "C2", "stage": "fine", "s_val": 20.0}, "_detail": {},
       "_note": f"HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_c2.sh, "
                f"tag {tag}). FETCH_SIZE (KB) is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section); "
                f"WRITE_SIZE (KB) is taken as is. bench.py reports these as roofline.traffic."}
for call, v in per_call.items():
    if "fetch" in v and "write" in v:
        b = v["fetch"] * 1024 * 2 + v["write"] * 1024
        out[call] = b
        out["_detail"][call] = {"fetch_raw_kb": v["fetch"], "write_kb": v["write"], "hbm_bytes_corrected": b}
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
